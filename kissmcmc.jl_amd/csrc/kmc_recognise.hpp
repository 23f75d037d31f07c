// kmc_recognise.hpp -- the text matcher behind "a function body that is a sum over elements" (kmc_rtc.hip: recognise_separable), HOST-ONLY and
// free of HIP types: the C++ standard library is all it needs, so that it also builds with g++ -fsanitize=address,undefined into the fuzz harness
// tests/sanitize/recognise_fuzz.cpp (scripts/sanitize_cpu.sh).  It parses caller-supplied C text -- what stands in for the reference's arbitrary
// closure pdf(theta), src/samplers.jl:257 -- with hand-written string code and std::regex: exactly what a CPU sanitizer is for.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstring>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

namespace kmc_host {
struct SumForm {            // what a recognised body becomes
    int nacc = 0;           // accumulators (1 .. 4)
    bool pair = false;      // the loop reads x[i + 1] as well (runs to n - 1)
    std::string functor;    // source of `struct UserS` (term / pair / finish, or elem / finish): the per-element form the kernels instantiate
};

// ---- a function body that is a sum over elements ------------------------------------------------------------------------
// Recognised form (after comments are dropped; whitespace free):
//     [declarations not touching x]                       e.g.  const double w = p[1];
//     double ACC = 0[, ACC2 = 0 ...];                     the accumulators (one to four)
//     for (int I = 0; I < n; ++I)  STMT | { STMTS }       or  I + 1 < n  /  I < n - 1  when x[I + 1] is read
//     return EXPR;                                        any expression of ACC, n, p and the declarations
// where the loop's statements read the proposal only as x[I] (and x[I + 1]), change ACC only by `ACC += ...`, and contain no
// return / break / continue / goto / nested loop.  Then  logpdf = EXPR(ACC = sum over I of the loop body's increments), and the
// loop body is exactly a term (or pair) function of a TermPairDensity -- same operations per element, only the ORDER of the sum
// differs (lane-striped, like the menu densities); with several accumulators one pass over the elements feeds all the sums
// (SepDensityN).  Anything else -- early returns, x[j] with another index, a running sum read inside the loop --
// is not recognised and keeps running one walker per lane.  KMC_DEBUG=no-body-routing switches the recognition off.
namespace recognise_detail {
inline std::string strip_comments(const std::string& t)
{
    std::string o;
    for (size_t i = 0; i < t.size();) {
        if (t.compare(i, 2, "//") == 0) { while (i < t.size() && t[i] != '\n') ++i; }
        else if (t.compare(i, 2, "/*") == 0) { const size_t e = t.find("*/", i + 2); i = e == std::string::npos ? t.size() : e + 2; o += ' '; }
        else o += t[i++];
    }
    return o;
}
inline std::string squeeze(const std::string& t)          // every run of white space -> nothing (identifiers stay apart: see callers)
{
    std::string o;
    for (char c : t) if (!std::isspace((unsigned char)c)) o += c;
    return o;
}
inline bool has_word(const std::string& t, const std::string& w)
{
    return std::regex_search(t, std::regex("\\b" + w + "\\b"));
}
}  // namespace recognise_detail

inline bool recognise_sum_form(const std::string& text, SumForm* out)
{
    using namespace recognise_detail;
    if (text.size() > 4096) return false;            // (generated code: std::regex recurses per matched character -- stack -- and nobody writes a sum that long by hand)
    const std::string t = strip_comments(text);
    // one loop, of the canonical header
    static const std::regex head("for\\s*\\(\\s*int\\s+(\\w+)\\s*=\\s*0\\s*;([^;]*);([^)]*)\\)");
    std::smatch m;
    if (!std::regex_search(t, m, head)) return false;
    if (std::regex_search(m.suffix().first, t.end(), std::regex("\\b(for|while|do)\\b"))) return false;      // a second loop
    const std::string I = m[1].str(), cond = squeeze(m[2].str()), inc = squeeze(m[3].str());
    if (I == "x" || I == "n" || I == "p") return false;
    if (inc != "++" + I && inc != I + "++" && inc != I + "+=1") return false;
    bool to_n1;
    if (cond == I + "<n") to_n1 = false;
    else if (cond == I + "+1<n" || cond == I + "<n-1") to_n1 = true;
    else return false;
    // the loop's statement(s)
    size_t b = (size_t)(m.suffix().first - t.begin());
    while (b < t.size() && std::isspace((unsigned char)t[b])) ++b;
    std::string loop;
    size_t after;
    if (b < t.size() && t[b] == '{') {
        int depth = 0;
        size_t e = b;
        for (; e < t.size(); ++e) { if (t[e] == '{') ++depth; else if (t[e] == '}' && --depth == 0) break; }
        if (e >= t.size()) return false;
        loop = t.substr(b + 1, e - b - 1);
        after = e + 1;
    } else {
        const size_t e = t.find(';', b);
        if (e == std::string::npos) return false;
        loop = t.substr(b, e - b + 1);
        after = e + 1;
    }
    if (t.find('#') != std::string::npos) return false;                                              // a macro can hide anything from a text matcher
    if (std::regex_search(loop, std::regex("\\b(return|break|continue|goto|for|while|do|switch|static|thread_local|extern|volatile|register|asm|__asm__|__shared__|"
                                           "auto|struct|class|union|enum|typedef|using|new|delete|throw|try|catch|operator|template|decltype)\\b"))) return false;
    if (std::regex_search(loop, std::regex("(^|[^&])&([^&]|$)"))) return false;                      // an address taken (an element's, or a variable a callee may write)
    if (std::regex_search(loop, std::regex("(^|[^\\w\\]\\)\\s])\\s*\\["))) return false;               // a lambda
    // what follows the loop: exactly one return statement
    std::smatch r;
    const std::string tail = t.substr(after);
    if (!std::regex_match(tail, r, std::regex("\\s*return\\b([^;]*);\\s*"))) return false;
    const std::string ret = r[1].str();
    if (has_word(ret, "x") || has_word(ret, I)) return false;
    // the accumulators: every name the loop changes by += (at most four), each declared `double ACC = 0` in front of the loop
    std::vector<std::string> accs;
    {
        static const std::regex pluseq("\\b(\\w+)\\s*\\+=");
        for (auto it = std::sregex_iterator(loop.begin(), loop.end(), pluseq); it != std::sregex_iterator(); ++it) {
            const std::string name = (*it)[1].str();
            if (std::find(accs.begin(), accs.end(), name) == accs.end()) accs.push_back(name);
        }
        if (accs.empty() || accs.size() > 4) return false;
        for (const std::string& acc : accs) {
            if (acc == I || acc == "x" || acc == "n" || acc == "p") return false;
            // every other mention of an accumulator inside the loop would make the increments depend on a running sum
            const std::regex any_acc("\\b" + acc + "\\b"), inc_acc("\\b" + acc + "\\s*\\+=");
            const auto n_all = std::distance(std::sregex_iterator(loop.begin(), loop.end(), any_acc), std::sregex_iterator());
            const auto n_pe = std::distance(std::sregex_iterator(loop.begin(), loop.end(), inc_acc), std::sregex_iterator());
            if (n_all != n_pe) return false;
        }
        // compound assignments / increments of other variables: state carried between elements, or too clever for a text matcher
        if (std::regex_search(loop, std::regex("[-*/%&|^]=|<<=|>>=|\\+\\+|--"))) return false;
    }
    auto is_acc = [&](const std::string& name) { return std::find(accs.begin(), accs.end(), name) != accs.end(); };
    auto word_before = [&](size_t& j) {                       // the identifier ending just before loop[j] (white space skipped); j moves to its start
        while (j > 0 && std::isspace((unsigned char)loop[j - 1])) --j;
        const size_t e = j;
        while (j > 0 && (std::isalnum((unsigned char)loop[j - 1]) || loop[j - 1] == '_')) --j;
        return loop.substr(j, e - j);
    };
    // every `ACC +=` is a statement of its own (not a value inside an expression): at the start, after ; { } or `else`, or after the `)` of an if
    for (const std::string& acc : accs) {
        const std::regex inc_acc("\\b" + acc + "\\s*\\+=");
        for (auto it = std::sregex_iterator(loop.begin(), loop.end(), inc_acc); it != std::sregex_iterator(); ++it) {
            size_t j = (size_t)it->position(0);
            while (j > 0 && std::isspace((unsigned char)loop[j - 1])) --j;
            if (j == 0) continue;
            const char c = loop[j - 1];
            if (c == ';' || c == '{' || c == '}') continue;
            if (c == ')') {
                int depth = 0;
                size_t k = j;
                while (k > 0) { --k; if (loop[k] == ')') ++depth; else if (loop[k] == '(' && --depth == 0) break; }
                if (depth != 0) return false;
                if (word_before(k) != "if") return false;
                continue;
            }
            size_t k = j;
            if (word_before(k) != "else") return false;
        }
    }
    // every plain assignment is the initialiser of a declaration of a NEW scalar name: `double|int|... name = e` (anything else the loop
    // assigns to -- a variable from outside, the loop index, a static -- is state carried between elements)
    for (size_t k = 0; k < loop.size(); ++k) {
        if (loop[k] != '=') continue;
        if (k + 1 < loop.size() && loop[k + 1] == '=') { ++k; continue; }                              // ==
        if (k > 0 && std::strchr("=!<>+", loop[k - 1])) continue;                                     // == != <= >= += (the compound ones are gone)
        size_t j = k;
        const std::string name = word_before(j);
        if (name.empty() || std::isdigit((unsigned char)name[0])) return false;                       // a[i] = ..., *q = ..., (..) = ...
        if (name == I || name == "x" || name == "n" || name == "p" || is_acc(name)) return false;      // ... shadowed: x[I] would mean something else
        const std::string type = word_before(j);
        if (type != "double" && type != "int" && type != "float" && type != "bool" && type != "long" && type != "unsigned") return false;
    }
    // (declarations without an initialiser cannot shadow either)
    for (const std::string& name : {I, std::string("x"), std::string("n"), std::string("p")})
        if (std::regex_search(loop, std::regex("\\b(double|int|float|bool|long|unsigned|short|char|const)\\s+" + name + "\\b"))) return false;
    // the proposal is read as x[I] and x[I + 1] only
    bool reads_next = false;
    {
        static const std::regex xs("\\bx\\b\\s*(\\[([^\\]]*)\\])?");
        for (auto it = std::sregex_iterator(loop.begin(), loop.end(), xs); it != std::sregex_iterator(); ++it) {
            if (!(*it)[1].matched) return false;                              // x without an index (passed on, x + k, ...)
            const std::string idx = squeeze((*it)[2].str());
            if (idx == I) continue;
            if (idx == I + "+1" || idx == "1+" + I) { reads_next = true; continue; }
            return false;
        }
    }
    if (reads_next && !to_n1) return false;                                   // (would read past the row's end anyway)
    // the declarations in front of the loop: `[const] double|int a = e[, b = f];` only, the accumulators among them with the value 0
    const std::string pre = t.substr(0, (size_t)m.position(0));
    std::string prelude;                                                      // everything but the accumulators, for every generated function
    size_t acc_declared = 0;
    {
        size_t i = 0;
        while (i < pre.size()) {
            const size_t e = pre.find(';', i);
            const std::string st = pre.substr(i, (e == std::string::npos ? pre.size() : e) - i);
            i = e == std::string::npos ? pre.size() : e + 1;
            if (squeeze(st).empty()) continue;
            std::smatch d;
            if (!std::regex_match(st, d, std::regex("\\s*(const\\s+)?(double|int)\\s+(.*)"))) return false;
            if (has_word(st, "x") || st.find('{') != std::string::npos) return false;
            // split the declarators at top-level commas
            std::vector<std::string> decls;
            {
                std::string cur;
                int depth = 0;
                for (char c : d[3].str()) {
                    if (c == '(' || c == '[') ++depth;
                    if (c == ')' || c == ']') --depth;
                    if (c == ',' && depth == 0) { decls.push_back(cur); cur.clear(); } else cur += c;
                }
                decls.push_back(cur);
            }
            std::string kept;
            for (const std::string& dc : decls) {
                std::smatch a;
                if (std::regex_match(dc, a, std::regex("\\s*(\\w+)\\s*=\\s*(.*?)\\s*")) && is_acc(a[1].str())) {
                    if (d[2].str() != "double" || !std::regex_match(a[2].str(), std::regex("[-+]?0*\\.?0*"))) return false;    // an accumulator must start at 0
                    if (squeeze(a[2].str()).empty()) return false;
                    ++acc_declared;
                    continue;
                }
                for (const std::string& acc : accs) if (has_word(dc, acc)) return false;
                if (std::regex_match(dc, a, std::regex("\\s*(\\w+)\\b.*")) &&
                    std::regex_search(loop, std::regex("\\b" + a[1].str() + "\\s*(\\[[^\\]]*\\]\\s*)?=[^=]")))
                    return false;                                             // a declaration the loop assigns to: state carried between elements
                kept += (kept.empty() ? "" : ", ") + dc;
            }
            if (!kept.empty()) prelude += (d[1].matched ? "const " : "") + d[2].str() + " " + kept + "; ";
        }
    }
    if (acc_declared != accs.size()) return false;
    // the loop body as a function of (x[I] -> kmc_x, x[I + 1] -> kmc_y)
    std::string fn = std::regex_replace(loop, std::regex("\\bx\\s*\\[\\s*(" + I + "\\s*\\+\\s*1|1\\s*\\+\\s*" + I + ")\\s*\\]"), "kmc_y");
    fn = std::regex_replace(fn, std::regex("\\bx\\s*\\[\\s*" + I + "\\s*\\]"), "kmc_x");
    const size_t N = accs.size();
    std::ostringstream o;
    o << "namespace {\nstruct UserS {\n"
      << "  static constexpr bool kHasPair = " << (to_n1 ? "true" : "false") << ";\n";
    if (N == 1) {
        // one sum: the term / pair functor of a TermPairDensity, the return expression as its finish (SepDensity)
        const std::string& acc = accs[0];
        const std::string body_fn = "(void)kmc_x; (void)kmc_y; (void)" + I + "; (void)n; (void)p; " + prelude + "double " + acc + " = 0.0; { " + fn + " } return " + acc + ";";
        if (to_n1) {
            o << "  __device__ static double term(double, int, int, const double*) { return 0.0; }\n"
              << "  __device__ static double pair(double kmc_x, double kmc_y, int " << I << ", int n, const double* p) { " << body_fn << " }\n";
        } else {
            o << "  __device__ static double term(double kmc_x, int " << I << ", int n, const double* p) { const double kmc_y = 0.0; " << body_fn << " }\n"
              << "  __device__ static double pair(double, double, int, int, const double*) { return 0.0; }\n";
        }
        o << "  __device__ static double finish(double " << acc << ", int n, const double* p) { (void)n; (void)p; " << prelude << "return (" << ret << "); }\n};\n}\n";
    } else {
        // several sums: one pass over the elements adds to all of them (SepDensityN), the return expression sees them by name
        std::string zero, add, take;
        for (size_t q = 0; q < N; ++q) {
            zero += (q ? ", " : "double ") + accs[q] + " = 0.0";
            add += "kmc_acc[" + std::to_string(q) + "] += " + accs[q] + "; ";
            take += (q ? ", " : "const double ") + accs[q] + " = kmc_acc[" + std::to_string(q) + "]";
        }
        const std::string body_fn = "(void)kmc_x; (void)kmc_y; (void)" + I + "; (void)n; (void)p; " + prelude + zero + "; { " + fn + " } " + add;
        o << "  static constexpr int kNAcc = " << N << ";\n"
          << "  __device__ static void elem(double kmc_x, double kmc_y, int " << I << ", int n, const double* p, double (&kmc_acc)[" << N << "]) { " << body_fn << "}\n"
          << "  __device__ static double finish(const double (&kmc_acc)[" << N << "], int n, const double* p) { (void)n; (void)p; " << prelude << take << "; return (" << ret << "); }\n};\n}\n";
    }
    out->nacc = (int)N;
    out->pair = to_n1;
    out->functor = o.str();
    return true;
}
}  // namespace kmc_host

// Fuzz harness for the text matcher that stands in for the reference's arbitrary closure pdf(theta) (src/samplers.jl:257; product:
// kissmcmc.jl_amd/csrc/kmc_recognise.hpp, used by kmc_rtc.hip: recognise_separable).  Test infrastructure: built by scripts/sanitize_cpu.sh
// with g++ -fsanitize=address,undefined -- no HIP, no GPU.  stdin: records "<tag> <nbytes>\n<bytes>\n"; stdout: "<tag> <taken> <nacc> <pair>" per record.
#include <cstdio>
#include <iostream>
#include <string>

#include "../../kissmcmc.jl_amd/csrc/kmc_recognise.hpp"

int main()
{
    std::string tag;
    size_t n = 0, records = 0, taken = 0;
    while (std::cin >> tag >> n) {
        std::cin.get();                                   // the newline behind the header
        std::string body(n, '\0');
        std::cin.read(&body[0], (std::streamsize)n);
        if ((size_t)std::cin.gcount() != n) { std::fprintf(stderr, "short record\n"); return 2; }
        kmc_host::SumForm f;
        const bool ok = kmc_host::recognise_sum_form(body, &f);
        if (ok && (f.nacc < 1 || f.nacc > 4 || f.functor.find("struct UserS") == std::string::npos)) { std::fprintf(stderr, "inconsistent result\n"); return 3; }
        std::printf("%s %d %d %d\n", tag.c_str(), ok ? 1 : 0, f.nacc, f.pair ? 1 : 0);
        ++records;
        taken += ok;
    }
    std::fprintf(stderr, "recognise_fuzz: %zu records, %zu taken\n", records, taken);
    return 0;
}

// kmc_copy_kernels.hpp -- device kernels of kmc_copy.hip: the chain read-out in the reference's order (reference src/samplers.jl:219-221,
// :268-272) and the compaction of padded rows.  Not on the sampling path.
#pragma once
#include "kmc_kernels.hpp"

namespace kmc {

// Chain read-out in the reference's order, thetas[walker][sample] (src/samplers.jl:219-221, :268-272): K samples of the
// stored chain src [sample][walker][ld], walkers [w0, w0 + nw) -> dst [walker][..][nd] doubles with `dst_stride` elements
// from one walker to the next (K * nd: a dense piece; nsamples * nd: a block of a streamed chain).  Also the log-pdfs, as rows of
// one element.  A TILED transposition through LDS: a workgroup takes TW walkers x TK samples; it READS, sample by sample, the
// TW consecutive rows of that sample (contiguous in the chain: TW * ld elements) and WRITES, walker by walker, the run of TK samples
// (contiguous in the output: TK * nd doubles).  The first version read each walker's samples one by one -- rows nl * ld * 8 bytes
// apart, every 32-byte read in another page -- and moved 25 GB/s: 5.2 ms for the 131 MB chain of 4 096 walkers x 4 doubles x 1 000
// samples, 39 ms for 1 GB of 32-double rows, most of the drop-in call's wall time (profiles/r04_readout.txt).
// LDS: [TW][TK * nd + 1] doubles (the odd stride keeps a sample's TW rows off one bank).
// Rows longer than 4 096 elements go in column windows of that many (blockIdx.z), one row per tile.
struct ByWalkerTile { int32_t TW, TK, NC; uint32_t lds_bytes; };
__host__ __device__ inline ByWalkerTile by_walker_tile(int32_t nd)
{
    ByWalkerTile t;
    t.NC = nd <= 4096 ? nd : 4096;                                       // columns per tile
    t.TK = nd <= 200 ? 32 : (nd <= 4096 ? (6000 / nd > 0 ? 6000 / nd : 1) : 1);        // samples per tile: output runs of >= 256 bytes where a row allows
    const int64_t per = (int64_t)t.TK * t.NC + 1;
    int64_t tw = 7680 / per;                                             // within 60 KiB of LDS
    if (tw < 1) tw = 1;
    if (tw > 64) tw = 64;
    t.TW = (int32_t)tw;
    t.lds_bytes = (uint32_t)((int64_t)t.TW * per * 8);
    return t;
}
template <class T>
__global__ __launch_bounds__(256) void chain_by_walker(const T* __restrict__ src, double* __restrict__ dst, int64_t nl, int32_t ld,
                                                       int32_t nd, int64_t K, int64_t w0, int64_t nw, int64_t dst_stride, int32_t TW, int32_t TK, int32_t NC)
{
    extern __shared__ __attribute__((aligned(16))) double bw_tile[];
    const int64_t wt0 = (int64_t)blockIdx.x * TW, k0 = (int64_t)blockIdx.y * TK;
    const int c0 = (int)blockIdx.z * NC;
    const int tw = (int)(nw - wt0 < TW ? nw - wt0 : TW), tk = (int)(K - k0 < TK ? K - k0 : TK), nc = nd - c0 < NC ? nd - c0 : NC;
    if (tw <= 0 || tk <= 0 || nc <= 0) return;
    const int per = TK * NC + 1;
    if (NC == nd) {
        // in: sample k of the tile = tw * ld consecutive elements (the pad column of an odd ndim among them: skipped)
        const int row_in = tw * ld;
        for (int i = (int)threadIdx.x; i < tk * row_in; i += 256) {
            const int k = i / row_in, r = i - k * row_in;
            const int w = r / ld, c = r - w * ld;
            if (c < nd) bw_tile[w * per + k * nd + c] = (double)src[((k0 + k) * nl + (w0 + wt0 + w)) * ld + c];
        }
        __syncthreads();
        // out: walker w of the tile = tk * nd consecutive doubles
        const int run = tk * nd;
        for (int i = (int)threadIdx.x; i < tw * run; i += 256) {
            const int w = i / run, e = i - w * run;
            dst[(wt0 + w) * dst_stride + k0 * nd + e] = bw_tile[w * per + e];
        }
    } else {
        // a column window of a long row (TW = TK = 1): straight through
        const T* in = src + (k0 * nl + (w0 + wt0)) * ld + c0;
        double* out = dst + wt0 * dst_stride + k0 * nd + c0;
        for (int i = (int)threadIdx.x; i < nc; i += 256) out[i] = (double)in[i];
    }
}

// Padded rows [rows][ld] -> dense rows [rows][nd] (a streamed chain of odd ndim: compacted on the device, so that the copy to the
// host is one contiguous transfer).
__global__ __launch_bounds__(256) void rows_compact(const double* __restrict__ src, double* __restrict__ dst, int64_t rows, int32_t ld, int32_t nd)
{
    const int64_t n = rows * nd;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nd;
        dst[e] = src[r * ld + (e - r * nd)];
    }
}


}  // namespace kmc

// Probe: global -> LDS streaming by LDS-DMA (buffer_load_dwordx4 ... lds) the way the convolution kernels' K loop does it --
// every workgroup walks the K extent of its own 128 rows of a row-major [rows][rowbytes] matrix in steps, one barrier per
// step, the DMA DIST steps ahead -- with the row piece a step fetches being 64 bytes (half a cache line: what a 32-element
// fp16 K step of conv_igemm4 reads per row and plane) or 128 bytes (a whole line).  Same bytes per step and per workgroup in
// both forms.  Prints us per step and KB/us per CU.  No arithmetic: this is the memory side of the K loop alone.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_line_probe dma_line_probe.hip
// usage: dma_line_probe [rowbytes=2048] [wgs=512] [flush=1] [ncol=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0, 0);
#endif
}

// A step reads PIECE bytes of each of ROWS rows from each of four source matrices: the two planes of the A operand (rows of
// the workgroup's row tile, shared with the other `ncol - 1` column tiles of that row) and the two planes of the B operand (rows
// of its column tile, shared with every row tile) -- the access pattern of conv_igemm4's K loop.  ncol = 0: all four matrices
// private to the workgroup (pure streaming).
template <int PIECE, int NST, int ROWS>
__global__ __launch_bounds__(256, 2) void stream_kernel(const char* const src0, long long mat_bytes, int rowbytes, int steps,
                                                        long long* out, int ncol) {
    constexpr int NSRC = 4;
    constexpr int STAGE = NSRC * ROWS * PIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs[NSRC];
#pragma unroll
    for (int m = 0; m < NSRC; ++m) {
        long long row0 = (long long)blockIdx.x * ROWS;
        if (ncol > 0) row0 = (long long)(m < 2 ? blockIdx.x / ncol : blockIdx.x % ncol) * ROWS;
        rs[m] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src0 + m * mat_bytes + row0 * rowbytes), 0,
                                                  (unsigned)((long long)ROWS * rowbytes), 0x00020000);
    }
    // a 1-KiB DMA instruction: PIECE 64 -> 16 rows x 64 B (lane -> row lane >> 2, piece lane & 3);
    //                          PIECE 128 -> 8 rows x 128 B (lane -> row lane >> 3, piece lane & 7)
    constexpr int RPI = PIECE == 64 ? 16 : 8;          // rows per instruction
    constexpr int NI = ROWS / RPI / 4;                 // instructions per wave, source and step
    const int l_row = PIECE == 64 ? lane >> 2 : lane >> 3;
    const int l_piece = PIECE == 64 ? lane & 3 : lane & 7;
    unsigned voff[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) voff[q] = (unsigned)(((wave + 4 * q) * RPI + l_row) * rowbytes + l_piece * 16);
    int fstep = 0;
    auto issue = [&](char* stage) {
        const unsigned koff = (unsigned)((fstep * PIECE) % rowbytes);
#pragma unroll
        for (int m = 0; m < NSRC; ++m)
#pragma unroll
            for (int q = 0; q < NI; ++q) dma16(rs[m], stage + m * (STAGE / NSRC) + (wave + 4 * q) * 1024, voff[q] + koff);
        ++fstep;
    };
    constexpr int PER_STEP = NSRC * NI;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(smem + s * STAGE);
    int st_free = NST - 1, st_cur = 0;
    unsigned acc = 0;
    for (int k = 0; k < steps; ++k) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP * (NST - 2)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(smem + st_free * STAGE);
        acc += *reinterpret_cast<const unsigned*>(smem + st_cur * STAGE + tid * 16);  // touch the landed stage
        st_free = st_cur;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = acc;
    }
}

template <int PIECE, int NST, int ROWS>
double run(const char* src, long long mat_bytes, int rowbytes, int steps, int wgs, long long* dout, char* flush, size_t flush_bytes,
           bool do_flush, double* per_wg_us, int ncol) {
    const int lds = NST * 4 * ROWS * PIECE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<PIECE, NST, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        if (do_flush) CK(hipMemsetAsync(flush, rep, flush_bytes, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_kernel<PIECE, NST, ROWS>), dim3(wgs), dim3(256), lds, 0, src, mat_bytes, rowbytes, steps, dout, ncol);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<long long> h(2 * wgs);
    CK(hipMemcpy(h.data(), dout, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost));
    std::vector<double> t(wgs);
    for (int i = 0; i < wgs; ++i) t[i] = h[2 * i] / 100.0;
    std::sort(t.begin(), t.end());
    *per_wg_us = t[wgs / 2];
    return best * 1e3;
}

int main(int argc, char** argv) {
    const int rowbytes = argc > 1 ? atoi(argv[1]) : 2048;
    const int wgs = argc > 2 ? atoi(argv[2]) : 512;
    const bool flush = argc > 3 ? atoi(argv[3]) != 0 : true;
    const int ncol = argc > 4 ? atoi(argv[4]) : 4;
    const long long rows = 128ll * wgs;
    const long long mat_bytes = rows * rowbytes;
    char* src; long long* dout; char* fl;
    const size_t flush_bytes = 768ull << 20;
    CK(hipMalloc(&src, 4 * mat_bytes));
    CK(hipMemset(src, 1, 4 * mat_bytes));
    CK(hipMalloc(&dout, sizeof(long long) * 2 * wgs));
    CK(hipMalloc(&fl, flush_bytes));
    const int steps64 = rowbytes / 64, steps128 = rowbytes / 128;   // one pass over the K extent
    printf("rowbytes %d, %d workgroups, %s, %s\n", rowbytes, wgs, flush ? "caches flushed before the launch" : "operands cache-warm",
           ncol > 0 ? "A rows shared by ncol column tiles, B rows by all row tiles" : "private rows (pure streaming)");
    struct R { const char* name; double us; double wg; int steps; double kb; int per_cu; };
    std::vector<R> res;
    double us, wg;
    us = run<64, 2, 128>(src, mat_bytes, rowbytes, steps64, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"64-B pieces, 128 rows, 2 stages (64 KB) ", us, wg, steps64, 32, 2});
    us = run<64, 3, 128>(src, mat_bytes, rowbytes, steps64, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"64-B pieces, 128 rows, 3 stages (96 KB) ", us, wg, steps64, 32, 1});
    us = run<64, 4, 64>(src, mat_bytes, rowbytes, steps64, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"64-B pieces, 64 rows, 4 stages (64 KB)  ", us, wg, steps64, 16, 2});
    us = run<128, 2, 128>(src, mat_bytes, rowbytes, steps128, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"128-B lines, 128 rows, 2 stages (128 KB)", us, wg, steps128, 64, 1});
    us = run<128, 2, 64>(src, mat_bytes, rowbytes, steps128, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"128-B lines, 64 rows, 2 stages (64 KB)  ", us, wg, steps128, 32, 2});
    us = run<128, 3, 64>(src, mat_bytes, rowbytes, steps128, wgs, dout, fl, flush_bytes, flush, &wg, ncol);
    res.push_back({"128-B lines, 64 rows, 3 stages (96 KB)  ", us, wg, steps128, 32, 1});
    for (auto& r : res) {
        const double kb = r.kb * r.steps;  // per workgroup
        printf("  %s: launch %7.1f us, median workgroup %7.1f us = %.2f us per step of %2.0f KB; %d per CU: %6.1f KB/us per CU = %5.2f TB/s chip\n",
               r.name, r.us, r.wg, r.wg / r.steps, r.kb, r.per_cu, r.per_cu * kb / r.wg, r.per_cu * kb / r.wg * 256 / 1e3 / 1e3);
    }
    return 0;
}

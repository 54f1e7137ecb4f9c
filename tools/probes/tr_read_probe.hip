// Probe: ds_read_b64_tr_b16 as the operand fetch of v_mfma_f32_32x32x16_bf16 when the reduction index is the slow one
// in LDS ([k][channel] bf16 images, as in the weight-gradient GEMM).  D[m][n] = sum_k X[k][m] * Y[k][n], K = 16.
// Build: hipcc --offload-arch=gfx950 -O3 -o tr_read_probe tr_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ inline bf16x4 tr_read(const unsigned short* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
#else
    return bf16x4{};
#endif
}

// lane (r = lane & 31, h = lane >> 5) needs T[k = 8h + j][r], j = 0..7, from an image T[16][ld] (ld in elements)
__device__ inline bf16x8 frag(const unsigned short* T, int ld, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
    const unsigned short* a0 = T + (8 * h + q) * ld + 16 * g1 + 4 * pp;  // this lane's ADDRESS duty: row q, cols 4p..4p+3
    const bf16x4 lo = tr_read(a0);
    const bf16x4 hi = tr_read(a0 + 4 * ld);
    bf16x8 f;
    for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
    return f;
}

__global__ void probe(const unsigned short* X, const unsigned short* Y, float* D, unsigned short* dump) {
    __shared__ __attribute__((aligned(16))) unsigned short sx[16 * 32], sy[16 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) { sx[i] = X[i]; sy[i] = Y[i]; }
    __syncthreads();
    const bf16x8 a = frag(sx, 32, lane), b = frag(sy, 32, lane);
    for (int j = 0; j < 8; ++j) dump[lane * 8 + j] = __builtin_bit_cast(unsigned short, a[j]);
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    const int r = lane & 31, h = lane >> 5;
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    std::vector<unsigned short> X(512), Y(512), dump(512);
    for (int k = 0; k < 16; ++k)
        for (int c = 0; c < 32; ++c) {
            X[k * 32 + c] = f2bf((float)(k * 32 + c));           // exactly representable up to 256; coarser above: fine
            Y[k * 32 + c] = f2bf((float)((rand() % 17) - 8));
        }
    unsigned short *dX, *dY, *dd; float* dD;
    hipMalloc(&dX, 1024); hipMalloc(&dY, 1024); hipMalloc(&dd, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dY, Y.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dX, dY, dD, dd);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(dump.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad_frag = 0, bad_d = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j)
            if (dump[lane * 8 + j] != X[(8 * (lane >> 5) + j) * 32 + (lane & 31)]) ++bad_frag;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += (double)bf2f(X[k * 32 + m]) * bf2f(Y[k * 32 + n]);
            if (D[m * 32 + n] != (float)s) ++bad_d;
        }
    printf("fragment mismatches %d / 512, product mismatches %d / 1024\n", bad_frag, bad_d);
    if (bad_frag) {
        for (int lane = 0; lane < 64; lane += 9) {
            printf("lane %2d got:", lane);
            for (int j = 0; j < 8; ++j) printf(" %g", bf2f(dump[lane * 8 + j]));
            printf("\n");
        }
    }
    return bad_frag || bad_d;
}

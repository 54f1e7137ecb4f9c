// Probe: is fp32 GEMM emulated by 3-way bf16 splitting (6 products on v_mfma_f32_32x32x16_bf16, fp32 accumulate) as
// accurate as the fp32 MFMA (v_mfma_f32_32x32x2_f32)?  One wave computes a 32x32 tile over K; both are compared with an
// fp64 host result.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_split_probe mfma_split_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    float r = x - (float)h;
    m = (__bf16)r;
    float r2 = r - (float)m;
    l = (__bf16)r2;
}

// A [32][K] row-major, B [K][32] row-major, C [32][32]
__global__ void tile_f32(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

template <int NPROD>
__global__ void tile_bf16x3(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(A[r * K + k + 8 * h + j], x, y, z);
            ah[j] = x; am[j] = y; al[j] = z;
            split3(B[(k + 8 * h + j) * 32 + r], x, y, z);
            bh[j] = x; bm[j] = y; bl[j] = z;
        }
        // smallest terms first
        if (NPROD >= 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        }
        if (NPROD >= 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// separate accumulators per magnitude class, summed at the end (small + (mid + big))
__global__ void tile_bf16x3_sep(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0};
    for (int k = 0; k < K; k += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(A[r * K + k + 8 * h + j], x, y, z);
            ah[j] = x; am[j] = y; al[j] = z;
            split3(B[(k + 8 * h + j) * 32 + r], x, y, z);
            bh[j] = x; bm[j] = y; bl[j] = z;
        }
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a2, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a2, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a2, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a0, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = a0[i] + (a1[i] + a2[i]);
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

int main() {
    const int Ks[] = {64, 576, 4608, 18432};
    for (int mode = 0; mode < 3; ++mode) {  // 0: N(0,1) x N(0,1); 1: relu-like A (>=0), small B; 2: wide dynamic range
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(K * 32);
            for (auto& v : A) {
                double x = nrand();
                if (mode == 1) x = x > 0 ? x : 0;
                if (mode == 2) x *= std::pow(2.0, (int)(urand() * 40) - 20);
                v = (float)x;
            }
            for (auto& v : B) {
                double x = nrand() * (mode == 1 ? 0.02 : 1.0);
                if (mode == 2) x *= std::pow(2.0, (int)(urand() * 40) - 20);
                v = (float)x;
            }
            std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
            for (int i = 0; i < 32; ++i)
                for (int k = 0; k < K; ++k)
                    for (int j = 0; j < 32; ++j) {
                        double p = (double)A[i * K + k] * (double)B[k * 32 + j];
                        ref[i * 32 + j] += p;
                        mag[i * 32 + j] += std::fabs(p);
                    }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            const char* names[] = {"f32 mfma", "bf16x1", "bf16x3 (3 prod)", "bf16x3 (6 prod)", "bf16x3 (6, sep acc)", "host f32 seq"};
            for (int which = 0; which < 6; ++which) {
                std::vector<float> C(1024);
                if (which == 5) {
                    for (int i = 0; i < 32; ++i)
                        for (int j = 0; j < 32; ++j) {
                            float s = 0.f;
                            for (int k = 0; k < K; ++k) s = std::fmaf(A[i * K + k], B[k * 32 + j], s);
                            C[i * 32 + j] = s;
                        }
                } else {
                    if (which == 0) tile_f32<<<1, 64>>>(dA, dB, dC, K);
                    if (which == 1) tile_bf16x3<1><<<1, 64>>>(dA, dB, dC, K);
                    if (which == 2) tile_bf16x3<3><<<1, 64>>>(dA, dB, dC, K);
                    if (which == 3) tile_bf16x3<6><<<1, 64>>>(dA, dB, dC, K);
                    if (which == 4) tile_bf16x3_sep<<<1, 64>>>(dA, dB, dC, K);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                }
                double mx = 0, sq = 0;
                for (int i = 0; i < 1024; ++i) {
                    // error relative to the sum of |products| (the natural scale of a dot product's rounding error)
                    double e = std::fabs((double)C[i] - ref[i]) / mag[i];
                    mx = e > mx ? e : mx;
                    sq += e * e;
                }
                printf("mode %d K %6d %-22s max %.3e rms %.3e (x 2^-24: max %.2f rms %.3f)\n", mode, K, names[which], mx,
                       std::sqrt(sq / 1024), mx * 16777216.0, std::sqrt(sq / 1024) * 16777216.0);
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    return 0;
}

// Experiment: should the cached affine map of the fixed-model step (U_unc = K * [references; state terms], the only
// dense contraction on the c2 path) run on v_mfma_f64_16x16x4_f64 instead of vector FMAs?
//
//   variant A (what k_mpc<N, fixed-model> does): one wavefront per scenario, lane = row of K, K in LDS shared by the 4
//              waves of a workgroup, the scenario's data vector read as LDS broadcasts: one FMA + two LDS reads per term.
//   variant B: one wavefront per 16 scenarios, D[64 x 16] = K[64 x 108] * R[108 x 16] as 4 x 27 MFMAs; K is read from LDS
//              once per 16 scenarios, the data tile sits in LDS per wave.
//   variant C: a register-only loop of dependent-free MFMAs, to read off the instruction's issue rate on this chip.
//
// Build and run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_experiment tools/mfma_experiment.hip && /tmp/mfma_experiment
// Development tool; not part of libmpc_amd.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 50, NK = 2 * N + 5, NKP = 108, NR = 64;      // horizon, terms per row, padded terms, padded rows
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// K [NKP][NR] (term-major, zero padded), data [S][NKP] (scenario-major, zero padded), U [S][N]
__global__ __launch_bounds__(256) void k_vector(const double *K, const double *data, double *U, int S, int reps)
{
    extern __shared__ double lds[];
    double *sK = lds, *sD = lds + NKP * NR + (threadIdx.x >> 6) * NKP;
    for (int i = threadIdx.x; i < NKP * NR; i += 256) sK[i] = K[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int rep = 0; rep < reps; rep++)
        for (int s = wave; s < S; s += nw) {
            for (int k = lane; k < NKP; k += 64) sD[k] = data[(size_t)s * NKP + k];
            __builtin_amdgcn_wave_barrier();
            double a0 = 0.0, a1 = 0.0;
#pragma unroll 9
            for (int k = 0; k < NKP; k += 2) {
                a0 = fma(sK[k * NR + lane], sD[k], a0);
                a1 = fma(sK[(k + 1) * NR + lane], sD[k + 1], a1);
            }
            if (lane < N) U[(size_t)s * N + lane] = a0 + a1;
            __builtin_amdgcn_wave_barrier();
        }
}

__global__ __launch_bounds__(256) void k_mfma(const double *K, const double *data, double *U, int S, int reps)
{
    extern __shared__ double lds[];
    double *sK = lds, *sD = lds + NKP * NR + (threadIdx.x >> 6) * 16 * (NKP + 1);       // +1: rows of the data tile on different banks
    for (int i = threadIdx.x; i < NKP * NR; i += 256) sK[i] = K[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int li = lane & 15, lk = lane >> 4;
    for (int rep = 0; rep < reps; rep++)
        for (int s0 = wave * 16; s0 < S; s0 += nw * 16) {
            for (int e = lane; e < 16 * NKP; e += 64) {
                const int j = e / NKP, k = e % NKP;
                sD[j * (NKP + 1) + k] = (s0 + j < S) ? data[(size_t)(s0 + j) * NKP + k] : 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 3
            for (int kk = 0; kk < NKP / 4; kk++) {
                const double b = sD[li * (NKP + 1) + 4 * kk + lk];                  // B[k][j]: k = lane / 16, j = lane % 16
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double a = sK[(4 * kk + lk) * NR + 16 * t + li];          // A[i][k]: i = lane % 16, k = lane / 16
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {                                        // D[lane / 16 + 4 r][lane % 16] (checked against variant A)
                    const int row = 16 * t + lk + 4 * r;
                    if (row < N && s0 + li < S) U[(size_t)(s0 + li) * N + row] = acc[t][r];
                }
            __builtin_amdgcn_wave_barrier();
        }
}

__global__ __launch_bounds__(64) void k_mfma_rate(double *out, int n)
{
    double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

__global__ __launch_bounds__(64) void k_fma_rate(double *out, int n)
{
    double acc[16];
    for (int t = 0; t < 16; t++) acc[t] = threadIdx.x * 1e-3;
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1e-9;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int t = 0; t < 16; t++) acc[t] = fma(acc[t], a, b);
    }
    double s = 0;
    for (int t = 0; t < 16; t++) s += acc[t];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    const int S = 4096, reps = 200;
    std::vector<double> hK((size_t)NKP * NR, 0.0), hD((size_t)S * NKP, 0.0), uA((size_t)S * N), uB((size_t)S * N);
    srand(1);
    for (int k = 0; k < NK; k++) for (int r = 0; r < N; r++) hK[(size_t)k * NR + r] = rand() / (double)RAND_MAX - 0.5;
    for (int s = 0; s < S; s++) for (int k = 0; k < NK; k++) hD[(size_t)s * NKP + k] = rand() / (double)RAND_MAX - 0.5;
    double *dK, *dD, *dU, *dO;
    CHECK(hipMalloc(&dK, hK.size() * 8)); CHECK(hipMalloc(&dD, hD.size() * 8)); CHECK(hipMalloc(&dU, uA.size() * 8));
    CHECK(hipMalloc(&dO, 1024 * 64 * 8));
    CHECK(hipMemcpy(dK, hK.data(), hK.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dD, hD.data(), hD.size() * 8, hipMemcpyHostToDevice));
    const size_t ldsA = (size_t)(NKP * NR + 4 * NKP) * 8, ldsB = (size_t)(NKP * NR + 4 * 16 * (NKP + 1)) * 8;
    CHECK(hipFuncSetAttribute((const void *)k_vector, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    // A: 4096 scenarios = 1024 workgroups of 4 waves (as the product launches them); B: 256 waves = 64 workgroups
    for (int pass = 0; pass < 2; pass++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_vector, dim3(S / 4), dim3(256), ldsA, 0, dK, dD, dU, S, pass ? reps : 1);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) printf("A vector   : %8.3f ms for %d x %d scenario-maps  -> %.2f G maps/s, %.2f TFLOP/s useful\n", ms, reps, S, reps * (double)S / ms / 1e6,
                         2.0 * N * NK * reps * (double)S / ms / 1e9);
    }
    CHECK(hipMemcpy(uA.data(), dU, uA.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemset(dU, 0, uA.size() * 8));
    {
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma, dim3(S / 16 / 4), dim3(256), ldsB, 0, dK, dD, dU, S, pass ? reps : 1);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("B mfma f64 : %8.3f ms for %d x %d scenario-maps  -> %.2f G maps/s, %.2f TFLOP/s useful (64 workgroups: 16 scenarios per wave)\n", ms, reps, S,
                             reps * (double)S / ms / 1e6, 2.0 * N * NK * reps * (double)S / ms / 1e9);
        }
        CHECK(hipMemcpy(uB.data(), dU, uB.size() * 8, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t i = 0; i < uA.size(); i++) err = fmax(err, fabs(uA[i] - uB[i]));
        printf("max |A - B| = %.3e\n", err);
        if (!(err < 1e-12)) { printf("variants disagree\n"); return 1; }
    }
    // issue rates: one wave per SIMD on every CU (1024 waves), independent accumulators
    const int n = 20000;
    for (int pass = 0; pass < 2; pass++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma_rate, dim3(1024), dim3(64), 0, 0, dO, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) printf("C mfma rate: %8.3f ms, %d x 4 MFMA per wave, 1024 waves -> %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", ms, n,
                         1024.0 * n * 4 * 2048 / ms / 1e9, ms * 1e6 / (n * 4.0));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_fma_rate, dim3(1024), dim3(64), 0, 0, dO, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) printf("C fma rate : %8.3f ms, %d x 16 v_fma_f64 per wave, 1024 waves -> %.1f TFLOP/s, %.2f ns per FMA per SIMD\n", ms, n,
                         1024.0 * n * 16 * 128 / ms / 1e9, ms * 1e6 / (n * 16.0));
    }
    return 0;
}

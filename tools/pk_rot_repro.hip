// round 6, second attempt at a reproducer without product code: the arithmetic SHAPE of the failing routing kernel -- 16 lanes
// per row, 4 elements per lane, five row reductions, quotients, per-row scalars broadcast into packed multiplies -- compiled with the SLP
// vectoriser (hipcc default), run alone (reference bits) and then beside a kernel that holds two 200-register MFMA waves on every SIMD.
// Any bit that differs between the two runs is a wrong result: the kernel has no atomics and no data-dependent order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_rot_repro tools/pk_rot_repro.hip && ./pk_rot_repro [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rowsum16(float v)
{
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// out = x - rotate(x -> c) with the rotation trick's arithmetic (u, qh, w detached; arXiv 2410.06424), D = 64
__global__ void __launch_bounds__(256) victim(const float *x, const float *codes, const int *idx, float *out, int N)
{
    const int t = blockIdx.x * 256 + threadIdx.x, row = t >> 4, lane = t & 15;
    if (row >= N) return;
    const f32x4 e = *(const f32x4 *)(x + (size_t)row * 64 + lane * 4);
    const f32x4 c = *(const f32x4 *)(codes + (size_t)idx[row] * 64 + lane * 4);
    const float ne = sqrtf(rowsum16(e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w));
    const float nq = sqrtf(rowsum16(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w));
    const float de = fmaxf(ne, 1e-6f), dq = fmaxf(nq, 1e-6f);
    const f32x4 u = e / de, qh = c / dq, s = u + qh;
    const float ns = fmaxf(sqrtf(rowsum16(s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w)), 1e-6f);
    const f32x4 w = s / ns;
    const float a1 = rowsum16(e.x * w.x + e.y * w.y + e.z * w.z + e.w * w.w);
    const float a2 = rowsum16(e.x * u.x + e.y * u.y + e.z * u.z + e.w * u.w);
    const f32x4 r = (e - 2.f * a1 * w + 2.f * a2 * qh) * (nq / de);
    *(f32x4 *)(out + (size_t)row * 64 + lane * 4) = e - r;
}

// two waves of ~200 registers per SIMD (grid = 2 workgroups per CU), MFMAs back to back for `iters` rounds
__global__ void __launch_bounds__(256, 2) aggressor(float *sink, int iters)
{
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = (float)(threadIdx.x + r + k);
    f16x8 a[4], b;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) a[j][e] = (_Float16)(0.001f * (threadIdx.x + e + j));
    for (int e = 0; e < 8; ++e) b[e] = (_Float16)(0.002f * (e + 1));
    for (int i = 0; i < iters; ++i)
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 400, N = 3 * 65536 + 300, C = 256;
    float *hx = (float *)malloc((size_t)N * 64 * 4), *hc = (float *)malloc(C * 64 * 4);
    int *hi = (int *)malloc(N * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)N * 64; ++i) hx[i] = 3.f * ((float)rand() / RAND_MAX - 0.5f) * 3.46f;
    for (int i = 0; i < C * 64; ++i) hc[i] = ((float)rand() / RAND_MAX - 0.5f) * 3.46f;
    for (int i = 0; i < N; ++i) hi[i] = rand() % C;
    float *x, *c, *ref, *out, *sink; int *idx;
    hipMalloc(&x, (size_t)N * 256); hipMalloc(&c, C * 256); hipMalloc(&ref, (size_t)N * 256); hipMalloc(&out, (size_t)N * 256);
    hipMalloc(&idx, N * 4); hipMalloc(&sink, 4);
    hipMemcpy(x, hx, (size_t)N * 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, C * 256, hipMemcpyHostToDevice);
    hipMemcpy(idx, hi, N * 4, hipMemcpyHostToDevice);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    const unsigned blocks = (unsigned)(((size_t)N * 16 + 255) / 256);
    hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, sv, x, c, idx, ref, N);
    hipDeviceSynchronize();
    float *hr = (float *)malloc((size_t)N * 256), *ho = (float *)malloc((size_t)N * 256);
    hipMemcpy(hr, ref, (size_t)N * 256, hipMemcpyDeviceToHost);
    long long bad_alone = 0, bad_beside = 0, launches = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int r = 0; r < rounds; ++r) {
            hipMemsetAsync(out, 0xff, (size_t)N * 256, sv);
            hipStreamSynchronize(sv);
            if (mode) hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 0, sa, sink, 60000);      // ~ms of MFMAs on every SIMD
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, sv, x, c, idx, out, N);
            hipDeviceSynchronize();
            hipMemcpy(ho, out, (size_t)N * 256, hipMemcpyDeviceToHost);
            if (memcmp(ho, hr, (size_t)N * 256)) {
                long long n = 0; int first = -1;
                for (size_t i = 0; i < (size_t)N * 64; ++i) if (memcmp(ho + i, hr + i, 4)) { ++n; if (first < 0) first = (int)i; }
                if (mode) bad_beside += n; else bad_alone += n;
                if (n && (mode ? bad_beside : bad_alone) == n)
                    printf("first difference (%s): element %d of row %d (lane %d of its wave): %g vs %g\n", mode ? "beside" : "alone", first & 63, first >> 6,
                           ((first >> 6) & 3) * 16 + ((first & 63) >> 2), ho[first], hr[first]);
            }
            launches += 4;
        }
    printf("SLP-vectorised rotation kernel, %lld launches per mode: elements differing alone %lld, beside the MFMA kernel %lld\n", launches / 2, bad_alone, bad_beside);
    return 0;
}

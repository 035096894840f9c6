// atomic_bench.hip -- how fast can rows be scatter-added into a [C, D] fp32 table with global float atomics on MI355X?
// (the alternative to the counting sort + segmented sum of the EMA statistics: no second read of x)
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o /tmp/atomic_bench tools/atomic_bench.hip && /tmp/atomic_bench
// Variants: one table / one table per XCD (blockIdx % 8), agent vs workgroup scope, fp32 vs packed bf16 payload, rows per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SCOPE>
__device__ __forceinline__ void fadd(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE);
}

// one wave per row: lane l adds elements 4l .. 4l+3 (D = 256)
template <int SCOPE, bool PER_XCD, bool READ_X>
__global__ void __launch_bounds__(256) scatter_rows(const unsigned short *x, const int *idx, float *table, int64_t N, int C, int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float *tb = table + (PER_XCD ? (size_t)(blockIdx.x & 7) * C * 256 : 0);
    for (int r = 0; r < rows_per_wave; ++r) {
        const int64_t n = w * rows_per_wave + r;
        if (n >= N) return;
        const int c = idx[n];
        float v[4] = {1.f, 1.f, 1.f, 1.f};
        if (READ_X) {
            const uint2 u = *(const uint2 *)(x + n * 256 + lane * 4);
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
        float *p = tb + (size_t)c * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) fadd<SCOPE>(p + i, v[i]);
    }
}

// element-major variant: lane l adds elements l, l + 64, l + 128, l + 192 (each atomic instruction covers 256 contiguous bytes)
template <int SCOPE, bool PER_XCD>
__global__ void __launch_bounds__(256) scatter_rows_strided(const unsigned short *x, const int *idx, float *table, int64_t N, int C, int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float *tb = table + (PER_XCD ? (size_t)(blockIdx.x & 7) * C * 256 : 0);
    for (int r = 0; r < rows_per_wave; ++r) {
        const int64_t n = w * rows_per_wave + r;
        if (n >= N) return;
        const int c = idx[n];
        float *p = tb + (size_t)c * 256 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) fadd<SCOPE>(p + 64 * i, __uint_as_float((unsigned)x[n * 256 + lane + 64 * i] << 16));
    }
}

template <typename F>
static float time_it(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1000.f;
}

int main()
{
    const int64_t N = 1 << 20;
    const int C = 1024, D = 256;
    unsigned short *x; int *idx; float *table;
    CK(hipMalloc(&x, N * D * 2)); CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&table, (size_t)8 * C * D * 4));
    std::vector<int> h(N);
    srand(1);
    for (int64_t i = 0; i < N; ++i) h[i] = rand() % C;
    CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0x3c, N * D * 2));
    CK(hipMemset(table, 0, (size_t)8 * C * D * 4));
    for (int rpw : {1, 4, 16}) {
        const unsigned blocks = (unsigned)((N / rpw + 3) / 4);
#define RUN(name, ...) printf("%-64s rows/wave %2d : %8.1f us\n", name, rpw, time_it([&] { hipLaunchKernelGGL((__VA_ARGS__), dim3(blocks), dim3(256), 0, 0, x, idx, table, N, C, rpw); }, 5))
        RUN("agent scope, one table, 4 contiguous / lane, reads x", scatter_rows<__HIP_MEMORY_SCOPE_AGENT, false, true>);
        RUN("agent scope, one table, 4 contiguous / lane, no x", scatter_rows<__HIP_MEMORY_SCOPE_AGENT, false, false>);
        RUN("agent scope, table per XCD, 4 contiguous / lane, reads x", scatter_rows<__HIP_MEMORY_SCOPE_AGENT, true, true>);
        RUN("workgroup scope, table per XCD, 4 contiguous / lane, reads x", scatter_rows<__HIP_MEMORY_SCOPE_WORKGROUP, true, true>);
        RUN("workgroup scope, one table, 4 contiguous / lane, reads x", scatter_rows<__HIP_MEMORY_SCOPE_WORKGROUP, false, true>);
        RUN("agent scope, one table, strided lanes", scatter_rows_strided<__HIP_MEMORY_SCOPE_AGENT, false>);
        RUN("workgroup scope, table per XCD, strided lanes", scatter_rows_strided<__HIP_MEMORY_SCOPE_WORKGROUP, true>);
    }
    float s = 0; std::vector<float> t(16);
    CK(hipMemcpy(t.data(), table, 64, hipMemcpyDeviceToHost));
    for (float v : t) s += v;
    printf("checksum %g\n", s);
    return 0;
}

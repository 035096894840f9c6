// round 6: does `v_pk_mul_f32 v[0:1], v[2:3], v[0:1] op_sel_hi:[1,0]` -- a packed fp32 multiply whose HIGH half reads the
// register its LOW half overwrites -- return the architecturally defined result (hi = v3 * OLD v0) on gfx950 when the SIMD is shared with
// another kernel's MFMA waves?  No product code: a victim kernel that executes the instruction in a loop and checks both halves, an
// aggressor kernel that keeps the MFMA pipe and the VGPR read ports busy, run alone and side by side on two streams.
//   hipcc --offload-arch=gfx950 -O2 -o pk_inplace_repro tools/pk_inplace_repro.hip && ./pk_inplace_repro [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ void __launch_bounds__(256) victim(unsigned long long *bad, unsigned *lanes, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    unsigned long long nbad = 0;
    for (int i = 0; i < iters; ++i) {
        const float a = seed + 0.001f * (float)(lane + 1) + 0.37f * (float)(i & 1023);      // v0 (and the broadcast operand)
        const float b = 3.0f + (float)lane;                                                    // v1 (must NOT be read: op_sel_hi = 0)
        f32x2 d = {a, b};
        f32x2 s = {1.5f + 0.25f * (float)(i & 7), -2.25f - 0.5f * (float)(lane & 3)};          // v[2:3]
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[1,0]" : "+v"(d) : "v"(s));     // in place: dst == src1
        else { f32x2 o; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(o) : "v"(s), "v"(d)); d = o; }   // control: separate dst
        const bool ok = (d.x == s.x * a) && (d.y == s.y * a);
        if (!ok) { ++nbad; atomicOr(&lanes[lane >> 5], 1u << (lane & 31)); }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void __launch_bounds__(256) aggressor(float *sink, int iters)
{
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = (float)(threadIdx.x + r + k);
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (e + 1)); }
    for (int i = 0; i < iters; ++i)
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int FORM>
static unsigned long long run(int rounds, bool with_aggressor, unsigned *lanes_h)
{
    unsigned long long *bad; unsigned *lanes; float *sink;
    hipMalloc(&bad, 8); hipMalloc(&lanes, 8); hipMalloc(&sink, 4);
    hipMemset(bad, 0, 8); hipMemset(lanes, 0, 8);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    for (int r = 0; r < rounds; ++r) {
        if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, sa, sink, 4000);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(victim<FORM>, dim3(1024), dim3(256), 0, sv, bad, lanes, 2000, 1.0f + r);
        hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(lanes_h, lanes, 8, hipMemcpyDeviceToHost);
    hipStreamDestroy(sv); hipStreamDestroy(sa); hipFree(bad); hipFree(lanes); hipFree(sink);
    return h;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    const double per = 8.0 * 1024 * 256 * 2000;      // instruction instances (lanes) per round
    unsigned l[2];
    unsigned long long n;
    n = run<0>(rounds, false, l); printf("in-place form, alone on the chip:            %llu wrong of %.3g lane-results, lanes %08x%08x\n", n, per * rounds, l[1], l[0]);
    n = run<0>(rounds, true, l);  printf("in-place form, beside an MFMA-saturating kernel: %llu wrong of %.3g lane-results, lanes %08x%08x\n", n, per * rounds, l[1], l[0]);
    n = run<1>(rounds, true, l);  printf("separate destination, beside the MFMA kernel:  %llu wrong of %.3g lane-results, lanes %08x%08x\n", n, per * rounds, l[1], l[0]);
    return 0;
}

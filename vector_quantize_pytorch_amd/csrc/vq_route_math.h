// vq_route_math.h -- the value a VectorQuantize layer RETURNS for a row when gradients are routed to the input
// (vqp.py:1225-1233): straight-through `x + (q - x)` (vqp.py:282-283) or the rotation trick (vqp.py:287-318,
// arXiv:2410.06424 s4.2).  One definition for every kernel that needs it -- vq_route_kernel (one layer), vq_rvq_route_kernel
// (all stages of a residual loop) and the chain prologue of the screening kernel (vq_screen.hip) -- because ResidualVQ
// subtracts exactly this value from the residual (`residual - quantized.detach()`, rvq.py:524): the next stage's INDICES
// depend on its last bit, so the three kernels must agree bit for bit (built with -ffp-contract=off; IEEE division / sqrt).
//
// Layout contract: a row of D elements is spread over LPR consecutive lanes (LPR = 64: the wave; 16: one DPP row; 8: half a DPP
// row, the other half holds another tensor row); lane l of the row holds the NE elements 4 (LPR' k + l) + i, k < NE / 4,
// i < 4 in v[4 k + i], zero beyond D (LPR' = 16 for LPR = 8: a 32-element row in 8 lanes is the 16-lane layout whose upper
// eight lanes hold zeros, and adding their exact zero sums changes nothing).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "vqhip_internal.h"

// Sum over the 64 lanes, the same value in every lane.  DPP butterfly inside each row of 16 lanes (quad_perm, row_half_mirror,
// row_mirror: VALU only), then the four row sums through v_readlane -- no LDS traffic.
__device__ __forceinline__ float vq_wave_sum(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1, 0, 3, 2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2, 3, 0, 1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: every lane of a row holds the row's sum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Sum over the LPR lanes that share a row, the same value in each of them.  LPR = 16: one DPP row = one tensor row, four rows
// per wave, no scalar step at all.  LPR = 8: the first three butterfly steps (the fourth would add the other tensor row).
template <int LPR>
__device__ __forceinline__ float vq_row_sum(float v)
{
    static_assert(LPR == 64 || LPR == 16 || LPR == 8, "lanes per row");
    if (LPR == 64) return vq_wave_sum(v);
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});
    if (LPR == 16) v += dpp(v, std::integral_constant<int, 0x140>{});
    return v;
}

// bf16 rows: the reference runs rotate_to (vqp.py:287-318) on bf16 TENSORS, so every tensor op of it rounds its result to bf16
// (the reductions accumulate in fp32 and round once) -- and the rounded norms alone move the result by ~2 % of its magnitude against
// the fp32 formula.  With BF16 the frame and the forward value below apply those roundings op by op (the same sequence checked
// bit for bit against the live reference on 4096 x {16..256} rows; only the order of a row's fp32 sum is this kernel's own).
template <bool BF16>
__device__ __forceinline__ float vq_rb(float v)
{
    return BF16 ? (float)(__bf16)v : v;                       // v_cvt_pk_bf16_f32 (round to nearest even) + a shift
}

// two values rounded to bf16 by ONE v_cvt_pk_bf16_f32 (gfx950), unpacked by a shift and a mask
__device__ __forceinline__ void vq_rb2(float &a, float &b)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    const unsigned u = __builtin_bit_cast(unsigned, p);
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
}

// The rotation's frame for one row: u = e / |e|, qh = q / |q|, w = l2norm(u + qh), sc = |q| / |e| (all detached in the
// reference; safe_div's clamp at 1e-6, vqp.py:40-41; fp32: one reciprocal per row, then multiplies).
// BF16: every quotient below has bf16 operands (8 significant bits each).  Such a quotient is never closer than 2^-16.9 (relative)
// to the midpoint of two neighbouring bf16 values (exhaustive over the 128 x 128 significand pairs: the closest is 233 / 241), so
// bf16(a * rcp(b)) with v_rcp_f32's 1 ulp and the multiply's half ulp of fp32 is bit for bit bf16(fp32(a / b)): no IEEE division.
template <int NE, int LPR, bool BF16 = false>
__device__ __forceinline__ void vq_rot_frame(const float (&e)[NE], const float (&q)[NE], float (&u)[NE], float (&qh)[NE],
                                             float (&w)[NE], float &sc)
{
    static_assert(NE % 2 == 0, "elements per lane come in pairs");
    float se = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        se += e[k] * e[k];
        sq += q[k] * q[k];
    }
    const float ne = vq_rb<BF16>(sqrtf(vq_row_sum<LPR>(se))), nq = vq_rb<BF16>(sqrtf(vq_row_sum<LPR>(sq)));   // src.norm(), tgt.norm()
    const float de = vq_rb<BF16>(fmaxf(ne, 1e-6f)), dq = vq_rb<BF16>(fmaxf(nq, 1e-6f));                        // .clamp(min = eps)
    float st = 0.f;
    if (BF16) {
        const float ide = __builtin_amdgcn_rcpf(de), idq = __builtin_amdgcn_rcpf(dq);
#pragma unroll
        for (int k = 0; k < NE; k += 2) {
            u[k] = e[k] * ide; u[k + 1] = e[k + 1] * ide;             // safe_div(src, norm_src)
            vq_rb2(u[k], u[k + 1]);
            qh[k] = q[k] * idq; qh[k + 1] = q[k + 1] * idq;           // safe_div(tgt, norm_tgt)
            vq_rb2(qh[k], qh[k + 1]);
            w[k] = u[k] + qh[k]; w[k + 1] = u[k + 1] + qh[k + 1];     // u + q
            vq_rb2(w[k], w[k + 1]);
            st += w[k] * w[k];
            st += w[k + 1] * w[k + 1];
        }
        const float dn = vq_rb<true>(fmaxf(vq_rb<true>(sqrtf(vq_row_sum<LPR>(st))), 1e-6f));   // F.normalize: norm, clamp_min(eps)
        const float idn = __builtin_amdgcn_rcpf(dn);
#pragma unroll
        for (int k = 0; k < NE; k += 2) {
            w[k] *= idn; w[k + 1] *= idn;
            vq_rb2(w[k], w[k + 1]);
        }
        sc = vq_rb<true>(nq * ide);                           // safe_div(norm_tgt, norm_src)
        return;
    }
    const float ide = 1.f / de, idq = 1.f / dq;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        u[k] = e[k] * ide;
        qh[k] = q[k] * idq;
        w[k] = u[k] + qh[k];
        st += w[k] * w[k];
    }
    const float nt = fmaxf(sqrtf(vq_row_sum<LPR>(st)), 1e-6f);
    const float int_ = 1.f / nt;
#pragma unroll
    for (int k = 0; k < NE; ++k) w[k] = w[k] * int_;
    sc = nq / de;
}

// forward:  out = sc (e - 2 (e.w) w + 2 (e.u) qh)      (bf16: e @ w and e @ u are bmm results, their outer products with w / qh,
// the subtraction, the addition and the final scale each a bf16 tensor; the factor 2 is exact)
template <int NE, int LPR, bool BF16 = false>
__device__ __forceinline__ void vq_rot_fwd(const float (&e)[NE], const float (&u)[NE], const float (&qh)[NE], const float (&w)[NE],
                                           float sc, float (&t)[NE])
{
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) { a1 += e[k] * w[k]; a2 += e[k] * u[k]; }
    a1 = vq_rb<BF16>(vq_row_sum<LPR>(a1));
    a2 = vq_rb<BF16>(vq_row_sum<LPR>(a2));
    if (BF16) {
#pragma unroll
        for (int k = 0; k < NE; k += 2) {
            float p0 = a1 * w[k], p1 = a1 * w[k + 1];
            vq_rb2(p0, p1);
            float s0 = __builtin_fmaf(-2.f, p0, e[k]), s1 = __builtin_fmaf(-2.f, p1, e[k + 1]);   // (2 p exact: the same value as e - 2 p)
            vq_rb2(s0, s1);
            float r0 = a2 * qh[k], r1 = a2 * qh[k + 1];
            vq_rb2(r0, r1);
            s0 = __builtin_fmaf(2.f, r0, s0); s1 = __builtin_fmaf(2.f, r1, s1);
            vq_rb2(s0, s1);
            s0 *= sc; s1 *= sc;
            vq_rb2(s0, s1);
            t[k] = s0; t[k + 1] = s1;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) t[k] = (e[k] - 2.f * a1 * w[k] + 2.f * a2 * qh[k]) * sc;
}

// backward: grad_e = sc (g - 2 (g.w) w + 2 (g.qh) u)
template <int NE, int LPR>
__device__ __forceinline__ void vq_rot_bwd(const float (&g)[NE], const float (&u)[NE], const float (&qh)[NE], const float (&w)[NE],
                                           float sc, float (&t)[NE])
{
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) { a1 += g[k] * w[k]; a2 += g[k] * qh[k]; }
    a1 = vq_row_sum<LPR>(a1);
    a2 = vq_row_sum<LPR>(a2);
#pragma unroll
    for (int k = 0; k < NE; ++k) t[k] = sc * (g[k] - 2.f * a1 * w[k] + 2.f * a2 * u[k]);
}

// The routed forward value of one row in fp32: mode 1 straight-through, mode 2 rotation trick (anything else: the code row).
// BF16: the rows are bf16 tensors in the reference, whose `tgt - src` is itself a bf16 tensor (rounded) before it is added.
template <int NE, int LPR, bool BF16 = false>
__device__ __forceinline__ void vq_route_value(const float (&e)[NE], const float (&q)[NE], int mode, float (&t)[NE])
{
    if (mode == 2) {
        float u[NE], qh[NE], w[NE], sc;
        vq_rot_frame<NE, LPR, BF16>(e, q, u, qh, w, sc);
        vq_rot_fwd<NE, LPR, BF16>(e, u, qh, w, sc, t);
    } else if (mode == 1) {
#pragma unroll
        for (int k = 0; k < NE; ++k) t[k] = e[k] + (BF16 ? vq_bf16_bits_to_f32(vq_f32_to_bf16_rne(q[k] - e[k])) : (q[k] - e[k]));
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) t[k] = q[k];
    }
}

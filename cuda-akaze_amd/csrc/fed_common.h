// fed_common.h -- element traits, DPP wave shifts and the one-row FED update shared by the streaming kernels
// (kernels_fed.hip: FED steps only; kernels_fedsf.hip: low-pass + conductivity + FED steps in one pass).
#pragma once
#include "hak_internal.h"
#include <type_traits>

// The same streaming kernel serves both pipelines: V = float (akaze) and V = int (fastakaze, 16.16 fixed point,
// akazed.cu:3448-3470).  Integer arithmetic wraps (unsigned add / mul), so every regrouping used below for the float
// path (shared pair sums, shared flux products, tE + tW = P[x+1] - P[x]) is exact for the int path as well.
template <typename V> struct FedV;
template <> struct FedV<float> { using V4 = float4; };
template <> struct FedV<int> { using V4 = int4; };
__device__ __forceinline__ float4 mk4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
__device__ __forceinline__ int4 mk4(int a, int b, int c, int d) { return make_int4(a, b, c, d); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ int vadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int vsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int vmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
// L' from the flux sum:  float: fma(0.5*tau, sum, L) (akazed.cu:1263);  int: ((stepfac * (sum >> 16)) >> 16) + L (akazed.cu:3468-3469)
__device__ __forceinline__ float vstep(float f, float sum, float L) { return fmaf(f, sum, L); }
__device__ __forceinline__ int vstep(int f, int sum, int L) { return vadd(vmul(f, sum >> 16) >> 16, L); }

template <typename V, int NS>
struct FedFacs { V f[NS]; };

// bound_ctrl = 1 with a zero `old`: the end lane (no source) reads 0 -- lanes 0 and 63 are strip margin -- and the
// compiler needs no copy of `v` to seed the destination (an `old = v` shift costs one extra v_mov each)
__device__ __forceinline__ int hak_dpp_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i-1
__device__ __forceinline__ int hak_dpp_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }   // lane i <- lane i+1
__device__ __forceinline__ float wave_shr1(float v) { return __int_as_float(hak_dpp_shr1(__float_as_int(v))); }
__device__ __forceinline__ float wave_shl1(float v) { return __int_as_float(hak_dpp_shl1(__float_as_int(v))); }
// Integer shifts are made opaque to the optimiser: ROCm 7.2's DPP-combine pass folded `a - shift(b)` into
// `v_sub_u32_dpp shift(b), a` (operands swapped, no v_subrev) in k_hessian_stream<int>: Lx came out with the sign of its
// second term flipped in every lane's first column.  The empty asm keeps the shift a plain v_mov_b32_dpp.
__device__ __forceinline__ int wave_shr1(int v)
{
    int r = hak_dpp_shr1(v);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int wave_shl1(int v)
{
    int r = hak_dpp_shl1(v);
    asm volatile("" : "+v"(r));
    return r;
}

// 16-byte store with the `nt` bit: planes that are written once and next read sparsely or much later should not displace
// the streams other kernels are reading from L2 / Infinity Cache (measured: Hessian -3 %, +3 % end to end)
typedef float hak_v4f __attribute__((ext_vector_type(4)));
typedef int hak_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hak_store_nt(float4* p, const float4 v)
{
    const hak_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<hak_v4f*>(p));
}
__device__ __forceinline__ void hak_store_nt(int4* p, const int4 v)
{
    const hak_v4i t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<hak_v4i*>(p));
}

// ---- predicated 16-byte stores without a branch.  A store inside `if (row in segment && lane owns its columns)` makes the
// compiler's s_waitcnt insertion assume it was NOT issued, so the wait for a prefetched row becomes vmcnt(<loads in flight>)
// and -- vmcnt retiring in order -- every row then also waits for all earlier stores of the wave to complete: the streaming
// kernels exposed the full store latency once per row.  A raw buffer store drops lanes whose offset is out of range in
// hardware, so the predicate moves into the offset (HAK_BUF_OOB) and the instruction is unconditional and countable.
typedef unsigned hak_v4u __attribute__((ext_vector_type(4)));
// out-of-range marker = num_records of every resource made below.  Offsets are SUMS of a lane part (column + plane offset)
// and a wave-uniform row part, either of which may carry the marker: valid sums stay below 2^30 (launchers check plane
// offset + plane size), one marker gives [2^30, 2^31), two give 2^31 -- all out of range, none wraps.
constexpr unsigned HAK_BUF_OOB = 0x40000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hak_buf_rsrc(void* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, HAK_BUF_OOB, 0x00020000);     // raw buffer, gfx9 dword-3 format bits
}
// voff: byte offset per lane (>= HAK_BUF_OOB = the lane writes nothing).  aux 2 = nt (see hak_store_nt).
// The scalar offset field is deliberately the constant 0: with an SGPR there, LLVM's hazard recognizer assumes that a
// following VALU write of the 128-bit store data needs no wait state (GCNHazardRecognizer::createsVALUHazard), but on
// gfx950 the next instruction did overwrite the data of the last four lanes of each 16-lane pass before the store had
// read them (lanes 12-15 / 28-31 / ... of one row wrong, only in some code shapes).  Plane offsets therefore go into voff.
__device__ __forceinline__ void hak_buf_store_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, const float4 v)
{
    const hak_v4u d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, 0, 2);
}
__device__ __forceinline__ void hak_buf_store_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, const int4 v)
{
    const hak_v4u d = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, 0, 2);
}

// 16-byte streaming load of a row piece.  HAK_NT_LOADS (A/B builds only) marks it non-temporal.
__device__ __forceinline__ float4 hak_load_stream(const float4* p)
{
#ifdef HAK_NT_LOADS
    const hak_v4f t = __builtin_nontemporal_load(reinterpret_cast<const hak_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ int4 hak_load_stream(const int4* p)
{
#ifdef HAK_NT_LOADS
    const hak_v4i t = __builtin_nontemporal_load(reinterpret_cast<const hak_v4i*>(p));
    return make_int4(t.x, t.y, t.z, t.w);
#else
    return *p;
#endif
}

constexpr int pmod(int a, int m) { return ((a % m) + m) % m; }

// horizontal pair sums of one g row as seen by a lane: h[j] = g[x0+j-1] + g[x0+j], j = 1..4 (h4 needs the right lane's g.x)
template <typename V> struct GHrow { V h1, h2, h3, h4; };
__device__ __forceinline__ float vneg(float a) { return -a; }
__device__ __forceinline__ int vneg(int a) { return (int)(0u - (unsigned)a); }

// One output row (4 px per lane) of one FED level from shared flux products.
//   horizontal: P[j] = (g[x0+j-1] + g[x0+j]) * (L[x0+j] - L[x0+j-1]);  term_E(x) = P[x+1], term_W(x) = -P[x] exactly, so
//               (tE + tW) = P[j+1] - P[j] bit for bit.  P[0] is the left lane's P[4] -- the same two operands in the same
//               order -- and arrives by ONE wave shift instead of being recomputed (shift + subtract + multiply).
//   vertical:   Q[r] = (g[r] + g[r+1]) * (L[r+1] - L[r]);  term_S(r) = Q[r], term_N(r) = -Q[r-1] exactly (IEEE addition
//               commutes, negating a factor negates the product), so  ((tE + tW) + tS) + tN = ((P[j+1] - P[j]) + Qn) - Qp.
//               Each Q row is formed once and serves the row above and the row below it.
// Reflect-101 (akazed.cu:1251-1254): x == 0: tW = tE -> P[0] := -P[1];  x == w-1: tE = tW -> P[4] := -P[3];  the callers
// do the same in y: row 0: Qp := -Qn, row h-1: Qn := -Qp.  Same value and same rounding as the reference expression
//   (f+fE)(LE-L) + (f+fW)(LW-L) + (f+fS)(LS-L) + (f+fN)(LN-L)  then  fma(stepfac, sum, L)    (akazed.cu:1259-1263)
// for both element types (integer arithmetic wraps, so the regroupings are exact there too).
template <bool XEDGE, typename V, typename V4>
__device__ __forceinline__ V4 fed_row(const V4 Lc, const GHrow<V>& gh, const V4 Qn, const V4 Qp, int x0, int w, V stepfac)
{
    const V Lr = wave_shl1(Lc.x);
    const V d1 = vsub(Lc.y, Lc.x), d2 = vsub(Lc.z, Lc.y), d3 = vsub(Lc.w, Lc.z), d4 = vsub(Lr, Lc.w);
    const V P1 = vmul(gh.h1, d1), P2 = vmul(gh.h2, d2), P3 = vmul(gh.h3, d3);
    V P4 = vmul(gh.h4, d4);
    if (XEDGE) P4 = x0 + 3 == w - 1 ? vneg(P3) : P4;         // borderAdd(x,1,w) = w-2 (w % 4 == 0: x == w-1 is the last component)
    V P0 = wave_shr1(P4);                                   // (executes with every lane active: not inside the select)
    if (XEDGE) P0 = x0 == 0 ? vneg(P1) : P0;                // abs(x-1) = 1
    V4 o;
    o.x = vstep(stepfac, vsub(vadd(vsub(P1, P0), Qn.x), Qp.x), Lc.x);
    o.y = vstep(stepfac, vsub(vadd(vsub(P2, P1), Qn.y), Qp.y), Lc.y);
    o.z = vstep(stepfac, vsub(vadd(vsub(P3, P2), Qn.z), Qp.z), Lc.z);
    o.w = vstep(stepfac, vsub(vadd(vsub(P4, P3), Qn.w), Qp.w), Lc.w);
    return o;
}
// Q[r] of one level: gv = g[r] + g[r+1], Ln = row r+1, Lc = row r
template <typename V, typename V4>
__device__ __forceinline__ V4 fed_q(const V4 gv, const V4 Ln, const V4 Lc)
{
    return mk4(vmul(gv.x, vsub(Ln.x, Lc.x)), vmul(gv.y, vsub(Ln.y, Lc.y)), vmul(gv.z, vsub(Ln.z, Lc.z)), vmul(gv.w, vsub(Ln.w, Lc.w)));
}
template <typename V4> __device__ __forceinline__ V4 vneg4(const V4 a) { return mk4(vneg(a.x), vneg(a.y), vneg(a.z), vneg(a.w)); }
template <typename V4> __device__ __forceinline__ V4 vsel4(const bool c, const V4 a, const V4 b)
{
    return mk4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}


// ---- sigma=1 low-pass taps and the conductivity, per element type (kernels_smoothflow.hip, kernels_fedsf.hip).
// int: every pass of the separable Gaussian ends in >> 16 (akazed.cu:2922-2985), the gradient energy is formed in wrapping
// int32 and the conductivity is kept as (int)(g * 65536 + 0.5f) (akazed.cu:3406-3445).
template <typename V> struct SfTaps { V k0, k1, k2; };
__device__ __forceinline__ float sf_conv(float c, float a1, float b1, float a2, float b2, const SfTaps<float>& t)
{
    float ws = c * t.k0;
    ws += t.k1 * (a1 + b1);
    ws += t.k2 * (a2 + b2);
    return ws;
}
__device__ __forceinline__ int sf_conv(int c, int a1, int b1, int a2, int b2, const SfTaps<int>& t)
{
    const unsigned ws = (unsigned)c * (unsigned)t.k0 + (unsigned)t.k1 * (unsigned)(a1 + b1) + (unsigned)t.k2 * (unsigned)(a2 + b2);
    return (int)ws >> 16;
}
__device__ __forceinline__ float sf_dif2(float dx, float dy, float ikc) { return ikc * (dx * dx + dy * dy); }
__device__ __forceinline__ float sf_dif2(int dx, int dy, float ikc)
{
    return (float)(int)((unsigned)dx * (unsigned)dx + (unsigned)dy * (unsigned)dy) * ikc;
}
__device__ __forceinline__ void sf_store_g(float* o, float g) { *o = g; }
__device__ __forceinline__ void sf_store_g(int* o, float g) { *o = (int)(g * 65536 + 0.5f); }

__device__ __forceinline__ float sf_g_value(float g) { return g; }
__device__ __forceinline__ int sf_g_value_int(float g) { return (int)(g * 65536 + 0.5f); }
template <typename V> __device__ __forceinline__ V sf_g_as(float g);
template <> __device__ __forceinline__ float sf_g_as<float>(float g) { return g; }
template <> __device__ __forceinline__ int sf_g_as<int>(float g) { return (int)(g * 65536 + 0.5f); }

// ---- 1 / d for the PM_G2 conductivity g = 1 / (1 + dif2).  hipcc's correctly rounded float division costs ~11 instructions
// (v_div_scale x2, v_rcp, four fmas, v_div_fmas, v_div_fixup).  For 1 <= d < 2^64 the three-instruction sequence below --
// v_rcp_f32 plus one Newton step with explicit fmas -- returns the identical bits: checked exhaustively over all 2^29
// floats of that range on gfx950 (tools/rcp_exhaustive.hip; the library's own copy is re-checked by hak_op_rcp_check in
// the GPU tests).  Callers must send anything else (NaN, inf, >= 2^64) through the IEEE division.
__device__ __forceinline__ float hak_rcp_newton(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, y0, 1.0f);
    return fmaf(e, y0, y0);
}

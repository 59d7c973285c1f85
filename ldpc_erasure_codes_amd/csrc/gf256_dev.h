// gf256_dev.h -- GF(2^8) arithmetic on the device (poly 0x171, see internal.h).
//
// The reference multiplies through 256x256 lookup tables (GF_mult_lookup(a+1,b+1),
// Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:45).  On CDNA4 a per-byte table gather is the wrong
// shape: in packet mode one wavefront multiplies a whole row of packed bytes by ONE coefficient, so the
// coefficient is wave-uniform and the product of 4 packed bytes is three v_perm_b32 byte-selects:
//
//      c*x = c*(x & 7)  ^  c*((x>>3 & 7) << 3)  ^  c*((x>>6) << 6)
//
// with the 8 + 8 + 4 partial products of the coefficient held in 5 scalar registers (loaded from a
// 8 KB __constant__ table by one s_load_dwordx8).  9 VALU ops per dword (5 index ops + 3 v_perm_b32 + 1 v_bitop3_b32), no LDS,
// no MFMA (a GF(256) product is a table/shift-xor operation, not a dense contraction).
//
// Scalar (S = 1) paths use log/antilog tables staged in LDS:  a*b = exp[log a + log b].
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldpc_amd {

struct alignas(16) U4 {
    uint32_t x, y, z, w;
};

// [c][0..1]: c*i for i=0..7 ; [c][2..3]: c*(i<<3) ; [c][4]: c*(i<<6), i=0..3 ; [5..7] unused
// Defined here: this header is included by exactly one device translation unit (kernels.hip), so no
// relocatable device code is needed.
__constant__ uint32_t c_mul3[256 * 8];
__constant__ uint8_t c_log[256];
__constant__ uint8_t c_exp[512];
__constant__ uint8_t c_inv[256];

struct MulTab {
    uint32_t t0, t1, t2, t3, t4;
};

// c must be wave-uniform for the loads to become scalar loads (callers pass readfirstlane'd values).
__device__ __forceinline__ MulTab load_multab(uint32_t c)
{
    const uint32_t *p = &c_mul3[c * 8];
    MulTab t;
    t.t0 = p[0]; t.t1 = p[1]; t.t2 = p[2]; t.t3 = p[3]; t.t4 = p[4];
    return t;
}

// v_perm_b32 D, S0, S1, SEL : byte i of D = byte SEL.byte[i] of the 64-bit value {S0,S1}
// (selector 0-3 -> S1, 4-7 -> S0).  __builtin_amdgcn_perm(S0, S1, SEL).
// a ^ b ^ c in ONE instruction: gfx950 has no v_xor3_b32, but v_bitop3_b32 evaluates any three-input boolean function
// (truth table 0x96 = parity); the compiler does not form it from two xors by itself.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// the three byte-select index words of a dword (they depend on the data only, not on the coefficient)
struct Sel3 {
    uint32_t s0, s1, s2;
};
__device__ __forceinline__ Sel3 gfsel(uint32_t x)
{
    return Sel3{x & 0x07070707u, (x >> 3) & 0x07070707u, (x >> 6) & 0x03030303u};
}
__device__ __forceinline__ uint32_t gfmul4_sel(const MulTab &t, const Sel3 &s)
{
    return xor3(__builtin_amdgcn_perm(t.t1, t.t0, s.s0), __builtin_amdgcn_perm(t.t3, t.t2, s.s1), __builtin_amdgcn_perm(t.t4, t.t4, s.s2));
}
// acc ^ c * x
__device__ __forceinline__ uint32_t gfmac4_sel(uint32_t acc, const MulTab &t, const Sel3 &s)
{
    return xor3(acc, __builtin_amdgcn_perm(t.t1, t.t0, s.s0), __builtin_amdgcn_perm(t.t3, t.t2, s.s1)) ^ __builtin_amdgcn_perm(t.t4, t.t4, s.s2);
}

__device__ __forceinline__ uint32_t gfmul4(const MulTab &t, uint32_t x)
{
    return gfmul4_sel(t, gfsel(x));
}

__device__ __forceinline__ U4 gfmul16(const MulTab &t, const U4 &v)
{
    U4 r;
    r.x = gfmul4(t, v.x); r.y = gfmul4(t, v.y); r.z = gfmul4(t, v.z); r.w = gfmul4(t, v.w);
    return r;
}

__device__ __forceinline__ void gfmac16(U4 &acc, const MulTab &t, const U4 &v)
{
    acc.x = gfmac4_sel(acc.x, t, gfsel(v.x)); acc.y = gfmac4_sel(acc.y, t, gfsel(v.y));
    acc.z = gfmac4_sel(acc.z, t, gfsel(v.z)); acc.w = gfmac4_sel(acc.w, t, gfsel(v.w));
}

// scalar product through log/antilog tables (table pointers may be LDS or constant memory)
__device__ __forceinline__ uint32_t gfmul_log(const uint8_t *lg, const uint8_t *ex, uint32_t a, uint32_t b)
{
    return (a && b) ? ex[lg[a] + lg[b]] : 0u;
}

}  // namespace ldpc_amd

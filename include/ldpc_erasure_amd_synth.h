/*
 * ldpc_erasure_amd_synth.h -- deterministic, counter-based synthetic-input generator shared by
 * the CPU oracle (C), the C/C++ host, the HIP kernels and the Python tests (re-implemented with
 * numpy uint64 arithmetic in ldpc_erasure_codes_amd/synth.py).
 *
 * The reference draws everything from Matlab's unseeded rand() (source symbols:
 * Matlab/ErasureCodes_NonBinaryLDPCSim.m:173, erasures: :191-198 and
 * Matlab/LDPCErasureCodes_MessagePassingAlgSim.m:183-188, GF(256) coefficients: the commented rule at
 * Matlab/ErasureCodes_NonBinaryLDPCSim.m:51-58) or from threefry on the FPGA
 * (OpenCL/device/ldpc_erasure_decoder_top.cl:74-110).  Neither stream is reproducible outside the
 * reference, so every input of this project is generated from (seed, stream, index) by the SplitMix64
 * finaliser below: any element can be produced independently on any device.
 *
 * Header-only, plain C99 / C++ / HIP.  No state.
 */
#ifndef LDPC_ERASURE_AMD_SYNTH_H
#define LDPC_ERASURE_AMD_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LDPC_SYNTH_FN static __host__ __device__ __forceinline__
#else
#define LDPC_SYNTH_FN static inline
#endif

/* stream ids */
#define LDPC_SYNTH_STREAM_COEF    1u /* GF(256) coefficient of the idx-th non-zero of H (1..255)   */
#define LDPC_SYNTH_STREAM_SOURCE  2u /* source byte: idx = (frame*k + sym)*S + lane                  */
#define LDPC_SYNTH_STREAM_ERASE   3u /* uniform erasure draw: idx = frame*n + sym                    */
#define LDPC_SYNTH_STREAM_BURST_E 4u /* Gilbert-Elliott: erasure draw for global symbol idx          */
#define LDPC_SYNTH_STREAM_BURST_S 5u /* Gilbert-Elliott: state-transition draw for global symbol idx */
#define LDPC_SYNTH_STREAM_RS      6u /* RS source bytes                                              */

LDPC_SYNTH_FN uint64_t ldpc_synth_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* 64 uniform bits for (seed, stream, idx) */
LDPC_SYNTH_FN uint64_t ldpc_synth_u64(uint64_t seed, uint32_t stream, uint64_t idx)
{
    uint64_t key = ldpc_synth_mix64(seed ^ ((uint64_t)stream * 0xD1342543DE82EF95ull));
    return ldpc_synth_mix64(key + idx);
}

/* uniform 32 bits */
LDPC_SYNTH_FN uint32_t ldpc_synth_u32(uint64_t seed, uint32_t stream, uint64_t idx)
{
    return (uint32_t)(ldpc_synth_u64(seed, stream, idx) >> 32);
}

/* uniform byte 0..255 (Matlab: floor(GF_SIZE*rand), ErasureCodes_NonBinaryLDPCSim.m:173) */
LDPC_SYNTH_FN uint8_t ldpc_synth_byte(uint64_t seed, uint32_t stream, uint64_t idx)
{
    return (uint8_t)(ldpc_synth_u64(seed, stream, idx) >> 56);
}

/* uniform non-zero GF(256) element 1..255 (Matlab: floor(255*rand)+1, ...NonBinaryLDPCSim.m:56) */
LDPC_SYNTH_FN uint8_t ldpc_synth_nonzero(uint64_t seed, uint32_t stream, uint64_t idx)
{
    return (uint8_t)(1u + (uint32_t)((ldpc_synth_u64(seed, stream, idx) >> 32) % 255u));
}

/* The reference's uniform draw rand(1) lies in the OPEN interval (0,1); ours is (u32 + 0.5) / 2^32. */
LDPC_SYNTH_FN double ldpc_synth_uniform(uint64_t seed, uint32_t stream, uint64_t idx)
{
    return ((double)ldpc_synth_u32(seed, stream, idx) + 0.5) * (1.0 / 4294967296.0);
}

/* Bernoulli(p) = the reference's "rand(1) <= p" (LDPCErasureCodes_MessagePassingAlgSim.m:184,
 * Bursty_Error_Channel_Model_Generator.m:28,32,38,42) on the draw above, as an integer compare:
 * (u + 0.5) / 2^32 <= p  <=>  u < floor(p * 2^32 + 0.5).  p <= 0 never fires, p >= 1 always does. */
LDPC_SYNTH_FN uint64_t ldpc_synth_threshold(double p)
{
    if (p <= 0.0) return 0;
    if (p >= 1.0) return 0x100000000ull;
    return (uint64_t)(p * 4294967296.0 + 0.5);
}

LDPC_SYNTH_FN int ldpc_synth_bernoulli(uint64_t seed, uint32_t stream, uint64_t idx, uint64_t thresh)
{
    return (uint64_t)ldpc_synth_u32(seed, stream, idx) < thresh;
}

/* ------------------------------------------------------------------------------------------------------
 * Threefry4x32-20 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11), written
 * from the published algorithm: 20 rounds of add / rotate / xor on four 32-bit words, the rotation schedule
 * repeating every 8 rounds, a key injection (plus round counter) every 4 rounds, key-schedule parity word
 * 0x1BD11BDA.  The FPGA source kernel draws its erasures from it (OpenCL/device/ldpc_erasure_decoder_top.cl:74-105,
 * vendored Random123 header OpenCL/device/threefry.h): key = {1, seed, 0, 0}, counter word 0 incremented BEFORE
 * every draw (so symbol g of the run uses counter g + 1), and a symbol is erased iff (out[0] & 0x3F) < PER_numerator.
 * ldpc_fpga_erased() is that rule, so ldpc_amd_data_in produces the FPGA's own erasure stream for a given seed. */
LDPC_SYNTH_FN uint32_t ldpc_rotl32(uint32_t x, unsigned r) { return (x << r) | (x >> (32u - r)); }

LDPC_SYNTH_FN void ldpc_threefry4x32_20(const uint32_t ctr[4], const uint32_t key[4], uint32_t out[4])
{
    const unsigned R0[8] = {10, 11, 13, 23, 6, 17, 25, 18};
    const unsigned R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
    uint32_t ks[5];
    uint32_t x0, x1, x2, x3;
    ks[4] = 0x1BD11BDAu;
    for (int i = 0; i < 4; i++) { ks[i] = key[i]; ks[4] ^= key[i]; }
    x0 = ctr[0] + ks[0]; x1 = ctr[1] + ks[1]; x2 = ctr[2] + ks[2]; x3 = ctr[3] + ks[3];
    for (unsigned r = 0; r < 20; r++) {
        if ((r & 1u) == 0) {
            x0 += x1; x1 = ldpc_rotl32(x1, R0[r & 7]); x1 ^= x0;
            x2 += x3; x3 = ldpc_rotl32(x3, R1[r & 7]); x3 ^= x2;
        } else {
            x0 += x3; x3 = ldpc_rotl32(x3, R0[r & 7]); x3 ^= x0;
            x2 += x1; x1 = ldpc_rotl32(x1, R1[r & 7]); x1 ^= x2;
        }
        if ((r & 3u) == 3u) {
            const unsigned s = (r >> 2) + 1;
            x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5];
            x3 += s;
        }
    }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}

/* erasure flag of the g-th symbol (0-based, frames concatenated) of an FPGA data_in run */
LDPC_SYNTH_FN int ldpc_fpga_erased(uint32_t seed, uint64_t g, int per_numerator_div_64)
{
    const uint32_t ctr[4] = {(uint32_t)(g + 1u), 0u, 0u, 0u};
    const uint32_t key[4] = {1u, seed, 0u, 0u};
    uint32_t o[4];
    ldpc_threefry4x32_20(ctr, key, o);
    return (int)(o[0] & 0x3Fu) < per_numerator_div_64;
}

#endif /* LDPC_ERASURE_AMD_SYNTH_H */

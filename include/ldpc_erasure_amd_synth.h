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

#endif /* LDPC_ERASURE_AMD_SYNTH_H */

/* ldpc_erasure_amd_wire.h -- host-side wire format of the FEC stream: the step before / after the decode path
 * (SURVEY.md 8(f) rank 4).  Plain C, no GPU involved; implemented in csrc/wire.cpp, same shared library.
 *
 * Reference:
 *   FEC header   OpenCL/device/ldpc_erasure_encoder_VITA_in_UDP_out.cl:112-114 (repair packets), :175-177 (source
 *                packets): one 32-bit word {class:8 | block:8 | symbol:16} repeated in both halves of a 64-bit word;
 *                parsed at OpenCL/device/ldpc_erasure_decoder_with_reordering_logic.cl:82-85.
 *   sender       ...VITA_in_UDP_out.cl:84-129,168-211: the k source packets of a block go out as they arrive, each
 *                behind its header, then the n-k repair packets; block number + 1 (8 bits on the wire) per block.
 *   receiver     ...reordering_logic.cl:44-141,214-243: two codeword buffers (current block, next block), packets of
 *                any other block are dropped, a received packet clears its symbol's erasure flag, the current block
 *                is handed to the decoder when (:139)
 *                    cur_cnt == n  ||  (cur_cnt > k + round(0.8 (n-k)) && next_cnt > 10)
 *                                  ||  (cur_cnt > k + round(0.2 (n-k)) && next_cnt > 100)
 *                and the buffers rotate (:214-243).
 * That receiver file is a draft that does not compile (`elseif`, unbalanced braces, :88,107,122); what is restated is
 * its evident intent, with three stated deviations: block numbers wrap modulo 256 when the buffers rotate (the draft
 * increments an int forever although the wire field has 8 bits), packets whose symbol number is >= n are dropped (the
 * draft would write out of bounds), and the closed block is returned to the caller -- who batches blocks for
 * ldpc_amd_decode_batch -- instead of being decoded on the spot.  Counters count packets, duplicates included, as in
 * the draft.
 */
#ifndef LDPC_ERASURE_AMD_WIRE_H
#define LDPC_ERASURE_AMD_WIRE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_AMD_FEC_HEADER_BYTES 8
#define LDPC_AMD_FEC_CLASS_LDPC 0x01 /* FECClassCode, ...VITA_in_UDP_out.cl:58 / ...reordering_logic.cl:42 */

uint64_t ldpc_amd_fec_header_pack(unsigned fec_class, unsigned block, unsigned symbol);
void ldpc_amd_fec_header_unpack(uint64_t word, unsigned *fec_class, unsigned *block, unsigned *symbol);

/* frames [nframes][n][S] (encoded codewords) -> packets [nframes * n][8 + S], in transmission order.  The header is
 * stored as a little-endian 64-bit word (the FPGA writes a ulong into the datagram channel).  Returns 0 / -1. */
int ldpc_amd_fec_packetize(const uint8_t *frames, long nframes, int n, int S, unsigned fec_class, unsigned block0,
                           uint8_t *packets);

typedef struct ldpc_amd_fec_rx ldpc_amd_fec_rx;
int ldpc_amd_fec_rx_create(int n, int k, int S, ldpc_amd_fec_rx **rx);
void ldpc_amd_fec_rx_destroy(ldpc_amd_fec_rx *rx);
/* One received packet (8 + S bytes).  Returns 1 when this packet closed the current block: its n x S payload plane
 * (erased symbols zero, the decoder kernels' assumption 2) and its n erasure flags are copied to sym_out / erased_out
 * and its wire block number to *block_out; 0 when nothing was closed; -1 on bad arguments. */
int ldpc_amd_fec_rx_push(ldpc_amd_fec_rx *rx, const uint8_t *packet, uint8_t *sym_out, uint8_t *erased_out, int *block_out);
/* The batch form a receive thread uses: consumes packets [0, npackets) of a contiguous array (stride 8 + S) until
 * max_blocks blocks have been closed; block i goes to sym_batch[i][n][S] / erased_batch[i][n] / blocks[i] -- the layout
 * ldpc_amd_decode_batch takes.  Returns the number of blocks closed, *consumed = packets used (so the caller can
 * decode the batch and continue from there); -1 on bad arguments. */
int ldpc_amd_fec_rx_push_many(ldpc_amd_fec_rx *rx, const uint8_t *packets, long npackets, uint8_t *sym_batch,
                              uint8_t *erased_batch, int *blocks, int max_blocks, long *consumed);
/* End of stream: closes the current block if it holds any packet (1) -- call until it returns 0. */
int ldpc_amd_fec_rx_flush(ldpc_amd_fec_rx *rx, uint8_t *sym_out, uint8_t *erased_out, int *block_out);
/* Packets dropped so far because their block was neither current nor next, or their symbol number was >= n. */
long ldpc_amd_fec_rx_dropped(const ldpc_amd_fec_rx *rx);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_ERASURE_AMD_WIRE_H */

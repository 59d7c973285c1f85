// wire.cpp -- FEC header, packetiser and two-buffer reassembler (host only).  Interface and reference citations:
// include/ldpc_erasure_amd_wire.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/ldpc_erasure_amd_wire.h"

extern "C" {

uint64_t ldpc_amd_fec_header_pack(unsigned fec_class, unsigned block, unsigned symbol)
{
    // dout = 0xffffffff & ((class & 0xff) << 24 | (block & 0xff) << 16 | (symbol & 0xffff)); both halves carry it
    const uint64_t dout = 0x00000000ffffffffull & (((uint64_t)(fec_class & 0xffu) << 24) | ((uint64_t)(block & 0xffu) << 16) | (uint64_t)(symbol & 0xffffu));
    return ((dout << 32) & 0xffffffff00000000ull) | (dout & 0x00000000ffffffffull);
}

void ldpc_amd_fec_header_unpack(uint64_t word, unsigned *fec_class, unsigned *block, unsigned *symbol)
{
    if (fec_class) *fec_class = (unsigned)((word >> 24) & 0xffu);
    if (block) *block = (unsigned)((word >> 16) & 0xffu);
    if (symbol) *symbol = (unsigned)(word & 0xffffu);
}

static inline void put_le64(uint8_t *p, uint64_t v)
{
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
static inline uint64_t get_le64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

int ldpc_amd_fec_packetize(const uint8_t *frames, long nframes, int n, int S, unsigned fec_class, unsigned block0, uint8_t *packets)
{
    if (!frames || !packets || nframes < 0 || n <= 0 || n > 65536 || S <= 0) return -1;
    const size_t plen = (size_t)LDPC_AMD_FEC_HEADER_BYTES + (size_t)S;
    for (long f = 0; f < nframes; f++)
        for (int j = 0; j < n; j++) {   // symbolNum 0..k-1 source, then repair_sym_ind k..n-1
            uint8_t *p = packets + ((size_t)f * n + j) * plen;
            put_le64(p, ldpc_amd_fec_header_pack(fec_class, block0 + (unsigned)f, (unsigned)j));
            memcpy(p + LDPC_AMD_FEC_HEADER_BYTES, frames + ((size_t)f * n + j) * S, (size_t)S);
        }
    return 0;
}

struct ldpc_amd_fec_rx {
    int n, k, S;
    int desired_parity_rx, min_parity_rx;       // :54-55
    std::vector<uint8_t> sym[2], er[2];          // codeword_first / codeword_second
    int is_first_codeword;                       // :49  (1: buffer 0 holds the current block)
    int cur_block_num, next_block_num;           // :50-51
    int cur_block_num_cnt, next_block_num_cnt;   // :52-53
    long dropped;
};

int ldpc_amd_fec_rx_create(int n, int k, int S, ldpc_amd_fec_rx **out)
{
    if (!out || n <= 0 || n > 65536 || k <= 0 || k >= n || S <= 0) return -1;
    ldpc_amd_fec_rx *rx = new (std::nothrow) ldpc_amd_fec_rx();
    if (!rx) return -1;
    rx->n = n; rx->k = k; rx->S = S;
    rx->desired_parity_rx = (int)lround((n - k) * 0.8);
    rx->min_parity_rx = (int)lround((n - k) * 0.2);
    for (int b = 0; b < 2; b++) {   // :62-71 all symbols erased until a packet produces them, payload zero
        rx->sym[b].assign((size_t)n * S, 0);
        rx->er[b].assign((size_t)n, 1);
    }
    rx->is_first_codeword = 1;
    rx->cur_block_num = rx->next_block_num = -1;
    rx->cur_block_num_cnt = rx->next_block_num_cnt = 0;
    rx->dropped = 0;
    *out = rx;
    return 0;
}

void ldpc_amd_fec_rx_destroy(ldpc_amd_fec_rx *rx) { delete rx; }

long ldpc_amd_fec_rx_dropped(const ldpc_amd_fec_rx *rx) { return rx ? rx->dropped : -1; }

// hand the current block out and rotate the buffers (:214-243)
static void close_current(ldpc_amd_fec_rx *rx, uint8_t *sym_out, uint8_t *erased_out, int *block_out)
{
    const int cb = rx->is_first_codeword ? 0 : 1;
    if (sym_out) memcpy(sym_out, rx->sym[cb].data(), rx->sym[cb].size());
    if (erased_out) memcpy(erased_out, rx->er[cb].data(), rx->er[cb].size());
    if (block_out) *block_out = rx->cur_block_num;
    rx->cur_block_num = rx->next_block_num;                   // :216
    rx->next_block_num = (rx->next_block_num + 1) & 0xff;     // :217, modulo the 8-bit wire field
    rx->cur_block_num_cnt = rx->next_block_num_cnt;           // :218
    rx->next_block_num_cnt = 0;                               // :219
    memset(rx->sym[cb].data(), 0, rx->sym[cb].size());        // :220-240
    memset(rx->er[cb].data(), 1, rx->er[cb].size());
    rx->is_first_codeword = rx->is_first_codeword ? 0 : 1;    // :241
}

int ldpc_amd_fec_rx_push(ldpc_amd_fec_rx *rx, const uint8_t *packet, uint8_t *sym_out, uint8_t *erased_out, int *block_out)
{
    if (!rx || !packet) return -1;
    unsigned cls, blockNum, symbolNum;
    ldpc_amd_fec_header_unpack(get_le64(packet), &cls, &blockNum, &symbolNum);   // :82-85
    (void)cls;
    if (rx->cur_block_num == -1 && rx->next_block_num == -1) {                   // :88-91
        rx->cur_block_num = (int)blockNum;
        rx->next_block_num = (rx->cur_block_num + 1) & 0xff;
    }
    const uint8_t *payload = packet + LDPC_AMD_FEC_HEADER_BYTES;
    if ((int)symbolNum >= rx->n) {
        rx->dropped++;
    } else if ((int)blockNum == rx->cur_block_num) {                             // :98-105, :120-127
        const int b = rx->is_first_codeword ? 0 : 1;
        memcpy(rx->sym[b].data() + (size_t)symbolNum * rx->S, payload, (size_t)rx->S);
        rx->er[b][symbolNum] = 0;
        rx->cur_block_num_cnt += 1;
    } else if ((int)blockNum == rx->next_block_num) {                            // :107-114, :129-136
        const int b = rx->is_first_codeword ? 1 : 0;
        memcpy(rx->sym[b].data() + (size_t)symbolNum * rx->S, payload, (size_t)rx->S);
        rx->er[b][symbolNum] = 0;
        rx->next_block_num_cnt += 1;
    } else {
        rx->dropped++;                                                           // "ignore (i.e. drop) all blockNum's that are not current or next"
    }
    // :139
    if (rx->cur_block_num_cnt == rx->n ||
        (rx->cur_block_num_cnt > rx->k + rx->desired_parity_rx && rx->next_block_num_cnt > 10) ||
        (rx->cur_block_num_cnt > rx->k + rx->min_parity_rx && rx->next_block_num_cnt > 100)) {
        close_current(rx, sym_out, erased_out, block_out);
        return 1;
    }
    return 0;
}

int ldpc_amd_fec_rx_push_many(ldpc_amd_fec_rx *rx, const uint8_t *packets, long npackets, uint8_t *sym_batch,
                              uint8_t *erased_batch, int *blocks, int max_blocks, long *consumed)
{
    if (!rx || !packets || npackets < 0 || !sym_batch || !erased_batch || max_blocks < 1) return -1;
    const size_t plen = (size_t)LDPC_AMD_FEC_HEADER_BYTES + (size_t)rx->S, fsz = (size_t)rx->n * rx->S;
    int closed = 0;
    long i = 0;
    for (; i < npackets && closed < max_blocks; i++) {
        int blk = -1;
        const int rc = ldpc_amd_fec_rx_push(rx, packets + (size_t)i * plen, sym_batch + (size_t)closed * fsz,
                                            erased_batch + (size_t)closed * rx->n, &blk);
        if (rc < 0) return -1;
        if (rc == 1) {
            if (blocks) blocks[closed] = blk;
            closed++;
        }
    }
    if (consumed) *consumed = i;
    return closed;
}

int ldpc_amd_fec_rx_flush(ldpc_amd_fec_rx *rx, uint8_t *sym_out, uint8_t *erased_out, int *block_out)
{
    if (!rx) return -1;
    if (rx->cur_block_num == -1 || (rx->cur_block_num_cnt == 0 && rx->next_block_num_cnt == 0)) return 0;
    close_current(rx, sym_out, erased_out, block_out);
    return 1;
}

}  // extern "C"

// Paged GQA decode for LARGE kv groups (65..128 query heads per kv head and workgroup) with head dims up to (288, 256): gqa_decode_wide.hip.
// The generic kernel (gqa_decode.hip) serves everything else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi_gqa_wide {

constexpr int kTile = 32;             // keys per tile of the general instance (tile_keys(): what a given call's work list counts in)
constexpr int kDKP = 288, kDVP = 256; // padded head dims of the one instance

struct Params {
    const uint16_t *q, *k, *v;
    uint16_t *out;
    const int32_t *seq_lens, *block_table;
    float *ws_o;      // [rows][kDVP] fp32 partial (unnormalised) outputs
    float *ws_ml;     // [rows][2]    softmax reference (scaled log2 domain), sum
    int batch, q_heads, kv_heads, group, page_size, bt_stride, num_splits, lk, lv;
    int64_t q_sb, q_sh, k_sblk, k_srow, k_sh, v_sblk, v_srow, v_sh, o_sb, o_sh;
    float sm_scale;
    const int32_t *plan;          // decode_plan.h work list in tiles of tile_keys() keys; null = uniform num_splits
    // Sequences in exactly TWO pieces finish between their two workgroups (as in mla_decode_wide8s.hip): piece s exports the rows of the
    // other piece's heads (64 .. 127 | 0 .. 63) written through, posts pair_tag in pair_flags[2 (b kv_heads + kvh) + s], waits (bounded) for
    // the partner's word and writes the outputs of its own 64 heads -- the merge kernel's sums in the merge kernel's order.  need_merge
    // (one word) receives pair_tag whenever a workgroup leaves partials for the merge launch.  null = every piece leaves its partials.
    uint64_t *pair_flags, *need_merge;
    uint64_t pair_tag;
    int pair_withhold;            // tests: piece 1 keeps its word back, piece 0 runs into the bounded wait
};

// applies to: 64 < group, lk <= 288, lv <= 256 (multiples of 8), power-of-two pages of >= 32 keys, row strides whose in-page offsets fit 32 bits
bool applies(int group, int lk, int lv, int page_size, int64_t k_sblk, int64_t k_srow, int64_t v_sblk, int64_t v_srow);
// keys per tile for this call: 64 when V is a column prefix of the K rows and pages hold >= 64 keys, else 32 (needs k, v, strides, lk, lv, page_size)
int tile_keys(const Params &p);
// units: (sequence, kv head, split) triples of the uniform form, or the work list's item bound
void launch(const Params &p, int dtype, long long units, hipStream_t st);

}  // namespace mi_gqa_wide

// dispatch.h — the host-side functions the three dispatch files of libfastq_hip.so share (context.hip: contexts, options,
// device helpers; scan_dispatch.hip: the record scan and the shard protocol; stats_dispatch.hip: the statistics calls).
#pragma once
#include "ctx.h"

fqh_status do_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out);
fqh_status do_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap, bool reuse_index = false);
void drop_index_if_overlaps(fqh_ctx *ctx, const void *d_dst, uint64_t bytes);
fqh_status emit_index(fqh_ctx *ctx, fqh_idx_record *dst, uint64_t cap);
void enqueue_fused_commit(fqh_ctx *ctx);
fqh_status ensure_full_index(fqh_ctx *ctx);
fqh_status fail(fqh_ctx *ctx, fqh_status s, const char *msg);
void free_line_buffers(fqh_ctx *ctx);
size_t lines_bytes(size_t n_tiles);
uint16_t *lines_in_use(fqh_ctx *ctx);
bool same_scan(const fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in);

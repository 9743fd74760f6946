// internal.hpp -- functions one translation unit of libbtr_pointnet2.so offers the others
// (not part of the C ABI).  The whole-backbone entry points (backbone.hip) sequence the
// whole-layer ones (sa_layer.hip) and need a few of their pieces with extra operands.
#pragma once
#include <cstdint>
#include <functional>

#include "common.hpp"

namespace btr {

// fps_bucket.hip: CUs a kernel of the step can land on (the device's minus the eight the next
// batch's large-scene FPS holds, minus what an overlapped collective keeps): grids sized as ONE
// round of resident workgroups -- the row chunks of the streaming / fused / weight-gradient
// kernels -- count these, not the device's, or the leftover workgroups wait for a second round
int grid_cus();
// KB of dynamic LDS the large-scene FPS launch holds at least (its min-dists live there; what is
// left over keeps the step's LDS-using workgroups off its CUs); BTR_FPS_LDS_KB, default 128, 96
// when WORLD_SIZE > 1.  fps_lds_kb(np): what a launch over np points asks for
int fps_lds_reserve_kb();
int fps_lds_kb(int np);

// graph_cache.hip: body(stream) -- a fixed sequence of launches on `stream`, determined entirely by
// what the caller hashed into `key` (hash_bytes over its descriptor and every argument) -- issued
// launch by launch the first time a key is seen, captured into a HIP graph the second time, and
// replayed from then on.  *how (optional): 0 issued, 1 captured + launched, 2 replayed.  Issued as
// always inside somebody else's capture, while the GEMM trace is on, or with BTR_GRAPHS=0.
int graph_run(uint64_t key, hipStream_t stream, const std::function<int(hipStream_t)> &body,
              int *how);
uint64_t hash_bytes(uint64_t h, const void *p, size_t n);
struct SideLane {
  static constexpr int kEvents = 5 * BTR_GF_MAX_DECODER_LAYERS + 2;
  hipStream_t s[2] = {};   // side streams (the second only when asked for)
  hipEvent_t ev[kEvents] = {};
};
// the library's side stream(s) of `main` (nstreams 1 or 2; NULL: none)
SideLane *side_lane(hipStream_t main, int nstreams);
bool graph_capturing();     // this host thread is inside graph_run's capture
bool graphs_enabled();
// sa_mlp.hip: btr_gemm_trace_begin() is in effect (its event pairs do not belong in a graph)
bool gemm_trace_active();

// sa_mlp.hip: out_bcn (B, C, N) [and out_cl (B*N, C)] = f(scale * y + shift) [+ add];
// add (optional): a (B, C, N)-shaped operand whose batch elements are add_bstride floats apart
// (a channel slab of a wider (B, C', N) tensor), added to out_bcn only.
int pm_out_add(int b, int n, int c, int ldy, const float *y, const float *scale,
               const float *shift, int relu, float *out_bcn, float *out_cl, const float *add,
               long long add_bstride, hipStream_t stream);

// Bounding box of a 64-point bucket of the FPS's spatial sort (two float4: lo.xyz, hi.xyz).  The
// large-scene FPS kernel knows every bucket's box (its pruning test) and leaves them in the dead
// counting-sort area of its workspace; the ball query over the same buckets
// (ball_query_bucket.hip) then skips its own box pass.  fps_boxes_note / _lookup: which FPS
// workspace (this host thread's most recent launches) holds boxes for (b, n), and where.
struct Box8 {
  float x0, y0, z0, p0, x1, y1, z1, p1;
};
// The contract between the two calls (btr_pointnet2.h documents it for callers): the boxes sit
// in the "dead" counting-sort area of the FPS workspace, so the whole workspace must stay
// untouched between the FPS and the query.  It is CHECKED on the device, not assumed: the FPS
// kernel stamps every box with its launch's epoch (p0, bit pattern) and with the box's own
// position (p1 = box_stamp_pos), the host note remembers the epoch, and every query workgroup
// compares all stamps while it builds the super-bucket boxes.  A mismatch anywhere in the scene
// -- the area was overwritten, partially restored, or filled by another launch than the noted one
// -- makes that workgroup rebuild the super boxes from the sorted points and skip the per-bucket
// cull: slower, still the exact result (tests/test_ops_gpu.py).
__host__ __device__ inline unsigned box_stamp_pos(int i) { return 0xb0c5b0c5u ^ (unsigned)i; }
void fps_boxes_note(const void *workspace, int b, int n, const Box8 *boxes,
                    unsigned epoch);   // boxes NULL: none
const Box8 *fps_boxes_lookup(const void *workspace, int b, int n, unsigned *epoch);

// sa_mlp.hip: BatchNorm finalisation inside the statistics GEMM (the last workgroup of a column
// block to arrive -- a ticket per 128 / 64-column block, zeroed by the call's first kernel -- turns
// the per-workgroup partial sums into scale / shift / mean / invstd and updates the running
// statistics: what btr_sa_bn_finalize does, in its summation order).  bnfin_arm() hands the
// description to the NEXT statistics GEMM launched on this host thread (btr_sa_gemm_nt, _rc,
// _poolfwd, btr_pm_gemm_nt); it returns false when the fused form is off (BTR_BN_TICKET=0, the
// f32-input GEMMs, more than BTR_BN_TICKET_MAX_ROWS = 2048 rows) and the caller launches the
// finalize kernel itself.  Measured on the FSB step (20 steps, same box): off 5.02 ms, <= 2048 rows
// 5.01, <= 8192 rows 5.20, <= 16384 rows 5.19, <= 70000 rows 5.67, all 6.16: the one workgroup
// that is left adds the 2 x (workgroups) x 128 partials of its column block alone (the finalize
// kernel spreads them over channels x 64 slices), and every workgroup pays a device-scope release
// behind a C tile it has just written -- a 5 us launch is cheaper from 128 workgroups up.
struct BnFin {
  unsigned *ticket;            // [column blocks] zeroed before the launch
  const float *gamma, *beta;
  float *scale, *shift, *mean, *invstd, *running_mean, *running_var;
  const float *rbias;          // see bn_finalize_bias
  int nbias;
  double count;
  float eps, momentum;
  int fence;                   // 1: full device-scope fences around the ticket (not the default: +9 - 12 us per GEMM, same results)
};
constexpr int kBnTickets = 8;  // tickets per statistics GEMM (column blocks of >= 64 channels, n <= 512)
// sa_mlp.hip: per-point first layer of a set-abstraction MLP (see ppfl_gather_add_kernel)
int ppfl_forward(int b, int n, int m, int s, int nl, int rows, float inv_radius, const float *xyz,
                 const float *new_xyz, const int *idx, const float *P, const float *w0x, float *y0,
                 float *relx, float *part, int grid, hipStream_t st);
// sa_layer.hip: one weight-preparation launch for the layer calls between _begin and _end (issued
// twice by the caller: a collecting pass before _launch, the real pass after it)
void prep_batch_begin();
void prep_batch_launch(hipStream_t st);
void prep_batch_end();
// decoder.hip: btr_decoder_layer_backward on channel-last rows (see the definition)
struct DecoderRowsOut {
  float *dres1, *dqp0, *dqp1, *dkp;
};
int decoder_layer_backward_rows(const btr_decoder_layer_t *d, const btr_decoder_plan_t *p,
                                const float *x_cl, const float *key_cl, const float *qpos_cl,
                                const float *kpos_cl, const float *dout_bcp, const float *g0,
                                const float *g1, const float *g2, void *saved, float *grads,
                                float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                                const DecoderRowsOut *out, void *scratch, btr_stream_t stream);
// decoder.hip: the same in two parts for two streams (see the definition), and the forward with
// the cross-attention's key / value rows computed by a separate call
enum { kDecoderBwdChain = 1, kDecoderBwdRest = 2, kDecoderBwdKey = 4 };
int decoder_layer_backward_parts(const btr_decoder_layer_t *d, const btr_decoder_plan_t *p,
                                 const float *x_cl, const float *key_cl, const float *qpos_cl,
                                 const float *kpos_cl, const float *dout_bcp, const float *g0,
                                 const float *g1, const float *g2, void *saved, float *grads,
                                 float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                                 const DecoderRowsOut *out, void *scratch, int parts,
                                 btr_stream_t stream);
bool decoder_kv_separable(const btr_decoder_layer_t *d, const btr_decoder_plan_t *p);
int decoder_layer_kv(const btr_decoder_layer_t *d, const btr_decoder_plan_t *p,
                     const float *key_cl, const float *kpos_cl, void *saved, btr_stream_t stream);
int decoder_layer_forward_ex(const btr_decoder_layer_t *d, const btr_decoder_plan_t *p,
                             const float *x_cl, const float *key_cl, const float *qpos_cl,
                             const float *kpos_cl, float *out_bcp, float *out_cl, void *saved,
                             void *scratch, int kv_ready, int parts, btr_stream_t stream);
enum { kDecoderFwdSelf = 1, kDecoderFwdRest = 2 };
// sa_layer.hip: btr_pm_chain_backward with the output gradient as channel-last rows a0 (+ a1)
// instead of dout (b, c, n), and / or the input gradient left as rows in dx_rows (leading
// dimension = the padded input width) instead of dx (b, c, n)
int pm_chain_backward_rows(const btr_pm_chain_t *d, const btr_pm_plan_t *p, const float *x_cl,
                           const float *dout, const float *a0, const float *a1, void *saved,
                           float *grads, float *dx, float *dx_rows, void *scratch,
                           btr_stream_t stream);
// sa_mlp.hip: rows[r][c] = a0[r][c] (+ a1[r][c]) for c < C, zero up to ldr; zero / colpart as
// pm_rows_zero (same tiles, same summation order)
int pm_rows_in(int b, int n, int c, int ldr, const float *a0, const float *a1, float *rows,
               float *zero, int nzero, float *colpart, hipStream_t stream);
// sa_mlp.hip: the small-M NT GEMM on the bf16 planes of (a sub-block of) a weight matrix
int pm_gemm_nt_planes(int rows, int n, int k, const float *a, int lda, const void *planes, int kp,
                      long long ps, float *c, int ldc, const float *bias, hipStream_t s);
bool bnfin_rows_ok(long long rows);   // would bnfin_arm() accept a GEMM over this many rows?
bool bnfin_arm(const BnFin &fin, long long rows);

// sa_mlp.hip: btr_sa_bn_finalize; rbias (nbias entries, may be NULL): running_mean += momentum * rbias
int bn_finalize_bias(int n, int nblk, double count, float eps, float momentum, const float *part,
                     const float *gamma, const float *beta, float *scale, float *shift,
                     float *mean, float *invstd, float *running_mean, float *running_var,
                     const float *rbias, int nbias, hipStream_t stream);

// sa_mlp.hip: btr_pm_rows that also clears `nzero` floats at `zero` (first workgroup) and, with
// colpart [b * cdiv(n, 64)][ldr], writes the column sums of every 64-row tile
int pm_rows_zero(int b, int n, int c, int ldr, const float *x, float *rows, float *zero,
                 int nzero, float *colpart, hipStream_t stream);

// sa_mlp.hip: btr_pm_gemm_nt (no prologue / bias / statistics) with the reduction split over
// `slices` workgroups per C tile; plane z of the partial products at parts + z * part_stride
int pm_splitk_slices(int rows, int n, int k);
int pm_gemm_nt_splitk(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                      float *parts, long long part_stride, int slices, hipStream_t stream);

// interpolate.hip: three_interpolate_grad through inverted index lists with caller-provided
// scratch; grad_out's batch elements are go_bstride floats apart (>= c * n).
size_t ti_grad_workspace_bytes(int b, int n, int m);
int ti_grad_lists(int b, int c, int n, int m, const float *grad_out, long long go_bstride,
                  const int *idx, const float *weight, float *grad_points, void *workspace,
                  size_t workspace_bytes, hipStream_t stream, int mode = 0 /* kScatter* */);

// ball_query.hip: measurement hook (btr_ball_query_time_next): the event pair the next ball
// query call of this host thread records around its launches
hipEvent_t *bq_call_events();

// sa_mlp.hip: between begin and flush the split-K reductions of the btr_sa_gemm_tn* calls of
// this host thread are collected and issued as ONE launch by the flush, on `stream` (all of
// them must have been issued on that stream, each with its own partials buffer)
void reduce_batch_begin();
void reduce_unpad_next(int kpad, int kout, int ldo = 0, int coff = 0);   // (ldo = 0: kout)
void reduce_batch_flush(hipStream_t stream);

// sa_mlp.hip: btr_sa_scatter / btr_sac_scatter split in two: the inverted neighbour lists only
// depend on the ball-query result, so a caller that has it early (btr_backbone_sampling) builds
// them ahead (mode 1) and the backward only reduces (mode 2); mode 0 = both, as the C entry points.
enum { kScatterBoth = 0, kScatterBuild = 1, kScatterReduce = 2 };
int sa_scatter_ex(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                  const float *dx0, const int *idx, float *dfeat_cl, float *dxyz,
                  float *dnew_xyz, void *workspace, size_t workspace_bytes, int mode,
                  hipStream_t stream);
int sac_scatter_ex(int b, int n, int m, int c, int ldx, int use_xyz, const float *dx0,
                   const int *cidx, const int *goff, float *dfeat_cl, void *workspace,
                   size_t workspace_bytes, int max_rows, int mode, hipStream_t stream);

// What btr_backbone_sampling prepares for one set-abstraction layer besides the ball query:
// the compact-row plan (NULL pointers: the layer is not compact / builds its own) and the
// filled scatter workspace of the backward (NULL: built in the backward).
struct SaGeom {
  int *goff = nullptr, *dims = nullptr, *cidx = nullptr, *bgrp = nullptr;
  float *bw = nullptr;
  void *scatter_ws = nullptr;
};
int sa_layer_forward_geom(const btr_sa_layer_t *d, const btr_sa_plan_t *plan, const float *xyz,
                          const float *new_xyz, const float *feats_cl, const int *idx,
                          float *out, float *out_cl, void *saved, void *scratch,
                          const SaGeom *geom, btr_stream_t stream);

// sa_layer.hip: btr_sa_layer_backward with an operand added to the feature gradient it writes
// (dfeat (b, c, n) = layer's own gradient + dfeat_add; dfeat_add's batch stride in floats).
int sa_layer_backward_add(const btr_sa_layer_t *d, const btr_sa_plan_t *plan, const int *idx,
                          const float *out, const float *dout, void *saved, float *grads,
                          float *dfeat, float *dxyz, float *dnew_xyz, void *scratch,
                          const float *dfeat_add, long long dfeat_add_bstride,
                          const SaGeom *geom, btr_stream_t stream);


// attention.hip: btr_attention_fwd / _bwd with out / dout strided like q (out[l][b][.] at
// out + l*o_sl + b*o_sb): the decoder layer keeps its rows batch-major
int attention_fwd_strided(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                          long long q_sb, const float *k, const float *v, long long kv_sl,
                          long long kv_sb, float *out, long long o_sl, long long o_sb,
                          float *lse, float scale, float dropout_p, unsigned long long seed,
                          const long long *step, btr_stream_t stream);
int attention_bwd_strided(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                          long long q_sb, const float *k, const float *v, long long kv_sl,
                          long long kv_sb, const float *out, const float *dout, long long o_sl,
                          long long o_sb, const float *lse, float *dsum, float *dq,
                          long long dq_sl, long long dq_sb, float *dk, float *dv,
                          long long dkv_sl, long long dkv_sb, float scale, float dropout_p,
                          unsigned long long seed, const long long *step, btr_stream_t stream);

enum { kAttnBwdQ = 1, kAttnBwdKV = 2 };
int attention_bwd_strided_parts(int lq, int lk, int b, int h, int d, const float *q,
                                long long q_sl, long long q_sb, const float *k, const float *v,
                                long long kv_sl, long long kv_sb, const float *out,
                                const float *dout, long long o_sl, long long o_sb,
                                const float *lse, float *dsum, float *dq, long long dq_sl,
                                long long dq_sb, float *dk, float *dv, long long dkv_sl,
                                long long dkv_sb, float scale, float dropout_p,
                                unsigned long long seed, const long long *step, int parts,
                                btr_stream_t stream);

}  // namespace btr

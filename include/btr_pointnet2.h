/*
 * btr_pointnet2.h -- C ABI of libbtr_pointnet2.so: the MI355X (gfx950) replacement for the
 * native layer of the reference's `pointnet2._ext` extension.
 *
 * What it replaces.  The reference's pybind module (src/bindings.cpp:11-24) exposes nine
 * functions whose C++ wrappers (src/{sampling,ball_query,group_points,interpolate}.cpp) check
 * the tensors, allocate the result and call one `*_kernel_wrapper(int sizes..., const float*,
 * ..., float*)` C function per op, which launches the CUDA kernel on the current stream.
 * Those nine `*_kernel_wrapper` functions are the native boundary; each entry point below is
 * the drop-in for one of them, with the same argument order and meaning plus
 *   - a trailing `btr_stream_t` (the reference reads at::cuda::getCurrentCUDAStream() itself),
 *   - an `int` status (0 = ok; the reference printf+exit(-1)s, include/cuda_utils.h:35-44).
 * All pointers are DEVICE pointers to contiguous row-major arrays; sizes are element counts.
 * No torch / ATen types cross this boundary.  Calls enqueue work on `stream` and return
 * without synchronising.  Inputs are never written.  Unlike the reference (whose wrappers
 * depend on torch::zeros-initialised outputs) every output element is fully defined by the
 * call itself: callers may pass uninitialised memory.
 *
 * File:line citations are relative to
 *   /root/reference/detection/Votenet/pointnet2/_ext_src/   (identical copy under GroupFree3D).
 */
#ifndef BTR_POINTNET2_H_
#define BTR_POINTNET2_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *btr_stream_t; /* a hipStream_t; NULL = the null stream */

#define BTR_OK 0
#define BTR_ERR_INVALID_ARGUMENT (-1) /* bad size / null pointer; nothing was launched */
/* positive values are hipError_t codes from the failed launch */

/* Library identification: returns BTR_ABI_VERSION the .so was built with. */
#define BTR_ABI_VERSION 1
int btr_abi_version(void);
/* Digest of the sources this library was built from (build.py build_id): measurements that are
 * quoted from committed profiles carry it, so a reader can tell whether they belong to it. */
const char *btr_build_id(void);

/* Rounding mode of the squared distance a*a + b*b + c*c this library was built with
 * (BTR_FMAD, csrc/common.hpp): every index the path returns (FPS arg-max, ball-query
 * membership, 3-NN order) is decided by it.  The reference's sources write the expression
 * out (sampling_gpu.cu:108-109, ball_query_gpu.cu:36-38, interpolate_gpu.cu:38) and leave the
 * contraction to nvcc (-O2, default --fmad=true, pointnet2/setup.py:22-25):
 *   1  libbtr_pointnet2.so        fma(c,c, fma(a,a, b*b))   nvcc/NVPTX contraction (default)
 *   0  libbtr_pointnet2_fmad0.so  ((a*a)+(b*b))+(c*c)       as written (--fmad=false build)
 *   2  libbtr_pointnet2_fmad2.so  fma(c,c, fma(b,b, a*a))   left-to-right chain
 * The three libraries export the same symbols and differ in nothing else. */
int btr_distance_mode(void);

/* Thread-local description of the last non-zero status returned on this thread ("" if none). */
const char *btr_last_error(void);

/* include/cuda_utils.h:20-24 `opt_n_threads`: the reference's block-size rule.  Exposed
 * because the FPS tie-break depends on it (see btr_furthest_point_sampling). */
int btr_opt_n_threads(int work_size);

/* -------------------------------------------------------------------------------------------
 * Replaces furthest_point_sampling_kernel_wrapper(b, n, m, dataset, temp, idxs)
 *   decl src/sampling.cpp:16-18, def src/sampling_gpu.cu:180-234, kernel :74-178.
 * dataset (b,n,3) f32 -> idxs (b,m) i32.  temp is the (b,n) f32 scratch of the reference
 * signature; it is overwritten (no pre-fill with 1e10 needed, src/sampling.cpp:78-80) and its
 * contents after the call are unspecified.
 * Bit-exact semantics: idxs[0]=0; points with x*x+y*y+z*z <= 1e-3 never compete; distances
 * in f32 with the rounding of this library's btr_distance_mode() (default 1: the FMA
 * contraction nvcc applies to the written expression; mode 0 = as written, no contraction);
 * the differences x-x0, y-y0, z-z0 are plain f32 subtractions in every mode, and the skip test
 * x*x+y*y+z*z uses the same mode; ties between equal maxima resolve as the
 * reference's 2^k-thread shared-memory tree does for block size opt_n_threads(n): smallest
 * (bitreverse(k mod bs), k).  m <= 0 is a no-op.
 * ------------------------------------------------------------------------------------------- */
int btr_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int *idxs, btr_stream_t stream);

/* Same, with the reference block size given explicitly (1,2,4,...,512): used by parity tests
 * to exercise the tie-break for every template instantiation of the reference kernel. */
int btr_furthest_point_sampling_bs(int b, int n, int m, const float *dataset, float *temp,
                                   int *idxs, int block_size, btr_stream_t stream);

/* Same with caller-provided scratch and an optional forced block size (0 = opt_n_threads(n)).
 * Large scenes (n > 4096) use a Morton-bucketed, bounding-box-pruned kernel that needs
 * btr_furthest_point_sampling_workspace_bytes() of scratch; the two entry points above take it
 * from hipMallocAsync.  With workspace == NULL large scenes fall back to the streaming kernel
 * (same results, slower), which uses `temp`. */
size_t btr_furthest_point_sampling_workspace_bytes(int b, int n, int m);
int btr_furthest_point_sampling_ws(int b, int n, int m, const float *dataset, float *temp,
                                   int *idxs, int block_size, void *workspace,
                                   size_t workspace_bytes, btr_stream_t stream);

/* Same result again, for a cloud the caller EXPECTS to be FPS-ordered -- levels 2-4 of a
 * sampling pyramid sample the points of the previous level in the order it chose them
 * (models/backbone_module.py:113-132 slices sa1_inds on exactly that assumption).  The answer is
 * then 0, 1, ..., m-1 unless a tie resolves differently at the smaller n; that hypothesis is
 * CHECKED in parallel (n * m independent distance tests, csrc/sampling.hip
 * fps_prefix_check_kernel) and the serial kernel only runs for scenes where it fails.  The
 * expectation is a performance hint, never a semantic input: any cloud gives the indices of
 * btr_furthest_point_sampling.  `scratch`: btr_fps_ordered_scratch_bytes() bytes (0 = shape not
 * covered: n > 4096, m > 2048 or m > n; the call then is btr_furthest_point_sampling_bs). */
size_t btr_fps_ordered_scratch_bytes(int b, int n, int m);
int btr_furthest_point_sampling_ordered(int b, int n, int m, const float *dataset, float *temp,
                                        int *idxs, int block_size, void *scratch,
                                        size_t scratch_bytes, btr_stream_t stream);

/* Measurement only (bench.py): the next large-scene btr_furthest_point_sampling* call of this
 * host thread records the two hipEvent_t around its sampling kernel alone (the spatial-sort
 * launches in front of it stay outside), on the call's stream.  NULL, NULL cancels. */
void btr_fps_time_next_kernel(void *start_event, void *stop_event);

/* Replaces gather_points_kernel_wrapper(b, c, n, npoints, points, idx, out)
 *   decl src/sampling.cpp:9-11, def src/sampling_gpu.cu:27-36, kernel :13-25.
 * out[b,c,j] = points[b,c,idx[b,j]];  points (b,c,n) f32, idx (b,npoints) i32. */
int btr_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                      float *out, btr_stream_t stream);

/* Replaces gather_points_grad_kernel_wrapper(b, c, n, npoints, grad_out, idx, grad_points)
 *   decl src/sampling.cpp:12-14, def src/sampling_gpu.cu:54-62, kernel :39-52.
 * grad_points (b,c,n) = scatter-add of grad_out (b,c,npoints); zeroed by this call. */
int btr_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points, btr_stream_t stream);

/* Replaces query_ball_point_kernel_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx)
 *   decl src/ball_query.cpp:9-11, def src/ball_query_gpu.cu:51-59, kernel :14-49.
 * new_xyz (b,m,3), xyz (b,n,3) f32 -> idx (b,m,nsample) i32: first nsample k (ascending) with
 * d2 < radius*radius (f32, strict), short rows padded with the first hit, empty rows zero. */
int btr_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, btr_stream_t stream);

/* Same with caller-provided scratch (what the torch shim uses: its caching allocator is
 * stream-ordered and free).  btr_ball_query itself takes the scratch from hipMallocAsync. */
size_t btr_ball_query_workspace_bytes(int b, int n, int m, int nsample);
int btr_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                      const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                      btr_stream_t stream);

/* What the library's one-round grids are sized for, and what the large-scene FPS launch holds
 * (read once per process from the environment; no GPU needed to ask):
 * btr_grid_cus(): CUs a kernel of the training step counts on = device CUs (256 without a
 *   device) - 8 for the next batch's FPS scenes - BTR_COMM_CUS for a collective that overlaps
 *   the backward (default 16 when WORLD_SIZE > 1 and BTR_DP=ddp, else 0); BTR_GRID_CUS overrides.
 * btr_fps_lds_reserve_kb(): KB of LDS the FPS launch holds at least on each of its CUs
 *   (BTR_FPS_LDS_KB; default 128, 96 when WORLD_SIZE > 1 so that an RCCL workgroup still fits
 *   beside a scene).  The running min-dists of the scene live in that LDS (4 B per point, curve
 *   order; what does not fit stays in the global workspace), the rest keeps other LDS-using
 *   workgroups off the CU.
 * btr_fps_lds_kb(points): KB a launch over scenes of `points` points asks for: on one GPU enough
 *   for all min-dists where they fit (159 KB: 40 704 points) and never below the floor; with
 *   BTR_FPS_LDS_KB or WORLD_SIZE > 1 exactly the floor.
 * btr_fps_set_lds_kb(kb): process-wide override (tests, A/B runs); kb < 0 restores the rules. */
int btr_grid_cus(void);
int btr_fps_lds_reserve_kb(void);
int btr_fps_lds_kb(int points);
void btr_fps_set_lds_kb(int kb);

/* Measurement only (bench.py): the next btr_ball_query_buckets call -- or btr_ball_query_ws call
 * on a scene of more than 4096 points -- of this host thread records the two hipEvent_t around
 * its launches, on the call's stream.  NULL, NULL cancels. */
void btr_ball_query_time_next(void *start_event, void *stop_event);

/* Replaces group_points_kernel_wrapper(b, c, n, npoints, nsample, points, idx, out)
 *   decl src/group_points.cpp:9-11, def src/group_points_gpu.cu:35-44, kernel :13-33.
 * out[b,c,j,k] = points[b,c,idx[b,j,k]]. */
int btr_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, btr_stream_t stream);

/* Replaces group_points_grad_kernel_wrapper(b, c, n, npoints, nsample, grad_out, idx,
 * grad_points)  decl src/group_points.cpp:13-15, def src/group_points_gpu.cu:71-80,
 * kernel :48-69.  grad_points (b,c,n) zeroed by this call, then scatter-added. */
int btr_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, btr_stream_t stream);

/* Replaces three_nn_kernel_wrapper(b, n, m, unknown, known, dist2, idx)
 *   decl src/interpolate.cpp:9-10, def src/interpolate_gpu.cu:66-73, kernel :14-64.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 SQUARED distances, idx (b,n,3) i32;
 * earliest index wins ties; m < 3 leaves +inf / index 0 in the unused slots. */
int btr_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, btr_stream_t stream);
/* The same launch also writing the feature-propagation blend weights the reference computes with
 * four torch ops on the result (pointnet2/pointnet2_modules.py:492-496):
 * weight (b, n, 3) = r / sum(r), r = 1 / (sqrt(dist2) + 1e-8). */
int btr_three_nn_weights(int b, int n, int m, const float *unknown, const float *known,
                         float *dist2, int *idx, float *weight, btr_stream_t stream);

/* Replaces three_interpolate_kernel_wrapper(b, c, m, n, points, idx, weight, out)
 *   decl src/interpolate.cpp:11-13, def src/interpolate_gpu.cu:108-117, kernel :77-106.
 * out[b,c,j] = points[b,c,i1]*w1 + points[b,c,i2]*w2 + points[b,c,i3]*w3 (left to right). */
int btr_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, btr_stream_t stream);

/* Replaces three_interpolate_grad_kernel_wrapper(b, c, n, m, grad_out, idx, weight,
 * grad_points)  decl src/interpolate.cpp:14-17, def src/interpolate_gpu.cu:150-159,
 * kernel :121-148.  grad_points (b,c,m) zeroed by this call, then scatter-added. */
int btr_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points,
                               btr_stream_t stream);

/* ===========================================================================================
 * Fused set-abstraction MLP (extension beyond the reference's nine native functions).
 *
 * The reference runs the per-group shared MLP + max-pool of PointnetSAModuleVotes
 * (pointnet2_modules.py:243-267: QueryAndGroup -> SharedMLP -> F.max_pool2d) as stock torch
 * ops on a materialised (B, 3+C, npoint, nsample) tensor.  The entry points below are the
 * building blocks of the MI355X replacement used by backtoreality_amd/pointnet2/fused_sa.py:
 * activations are channel-last with one row per (batch, centre, sample),
 * r = (b*m + j)*s + k, leading dimension a multiple of 4 floats.
 * =========================================================================================== */

/* x0[r][0..ldx): use_xyz ? (xyz[b,idx,:]-new_xyz[b,j,:])/radius_div : -, then the c feature
 * channels gathered from feats_cl (b,n,c) (channel-last), zero padded up to ldx. */
int btr_sa_gather(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                  const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                  float *x0, btr_stream_t stream);

/* c[rows][n] = f(a)[rows][k] . w[n][k]^T on the f32 MFMA; f = relu(pa[k]*y + pb[k]) when
 * pa/pb are given (the previous layer's BatchNorm+ReLU fused into the operand load);
 * part != NULL: per-workgroup column sums and sums of squares -> part[grid][2][n] with
 * grid = btr_sa_gemm_grid(rows) (BatchNorm statistics without a second pass). */
int btr_sa_gemm_grid(int rows);
int btr_sa_gemm_nt(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                   float *c, int ldc, const float *pa, const float *pb, float *part,
                   btr_stream_t stream);

/* part[nblk][2][n] -> per-channel scale/shift (gamma*invstd, beta-mean*scale), mean, invstd;
 * updates running_mean/var like nn.BatchNorm2d in training mode when they are non-NULL. */
int btr_sa_bn_finalize(int n, int nblk, double count, float eps, float momentum,
                       const float *part, const float *gamma, const float *beta, float *scale,
                       float *shift, float *mean, float *invstd, float *running_mean,
                       float *running_var, btr_stream_t stream);

/* out[b][c][j] = max_k relu(scale*y + shift) (+ channel-last copy, + arg-max k as u8). */
int btr_sa_pool(int b, int m, int s, int c, int ldy, const float *y, const float *scale,
                const float *shift, float *out, float *out_cl, unsigned char *arg,
                btr_stream_t stream);

/* Backward of pool+ReLU+BN: y is overwritten IN PLACE by dL/dy; part: [1024][2][c] scratch. */
int btr_sa_pool_bwd(int b, int m, int s, int c, int ldy, float *y, const float *dout,
                    const float *out, const unsigned char *arg, const float *mean,
                    const float *invstd, const float *scale, float *part, float *m1, float *m2,
                    float *dgamma, float *dbeta, btr_stream_t stream);

/* Backward of ReLU+BN of a hidden layer: g (dL/d activation) -> dL/dy IN PLACE.
 * c % 4 == 0, c <= 256; part: [1024][2][c] scratch. */
int btr_sa_bn_relu_bwd(long long rows, int c, int ld, float *g, const float *y,
                       const float *scale, const float *shift, const float *mean,
                       const float *invstd, float *part, float *m1, float *m2, float *dgamma,
                       float *dbeta, btr_stream_t stream);

/* dw[n][k] = sum_r g[r][n] * f(x[r][k]) (weight gradient); pw: [chunks][n][k] scratch with
 * chunks = btr_sa_gemm_tn_chunks(rows, n, k). */
int btr_sa_gemm_tn_chunks(int rows, int n, int k);
int btr_sa_gemm_tn(int rows, int n, int k, const float *g, int ldg, const float *x, int ldx,
                   const float *pa, const float *pb, float *pw, float *dw, btr_stream_t stream);

/* dL/dx0 back to the layer inputs: dfeat_cl (b,n,c) CHANNEL-LAST, dxyz (b,n,3), dnew_xyz
 * (b,m,3); each may be NULL and is fully written otherwise.  The neighbour lists are inverted
 * (integer atomics) and each point sums its rows: no float atomics. */
size_t btr_sa_scatter_workspace_bytes(int b, int n, int m, int s);
int btr_sa_scatter(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                   const float *dx0, const int *idx, float *dfeat_cl, float *dxyz,
                   float *dnew_xyz, void *workspace, size_t workspace_bytes,
                   btr_stream_t stream);

/* First-layer recompute.  A set-abstraction layer whose gathered input has <= 4 columns (SA1:
 * relative xyz + height; x0 [rows][4]) does not need its first pre-BN output y0 = x0 . w0^T
 * (rows x k floats) in HBM: btr_sa_gemm_nt with c == NULL only produces its BatchNorm
 * statistics, and the consumers rebuild y0 from the 16-byte input row:
 *   btr_sa_gemm_nt_rc      second layer forward,   A = relu(pa * y0 + pb)
 *   btr_sa_gemm_tn_rc      second layer weight gradient, X = relu(pa * y0 + pb)
 *   btr_sa_bn_relu_bwd_rc  first layer BN+ReLU backward fused with its weight gradient
 *                          (dw0 [k][4]); the dense dY0 is never written either.
 * Replaces, for that layer, the Conv2d / BatchNorm2d / ReLU forward and backward of
 * pytorch_utils.py:11-36 as used by pointnet2_modules.py:243-267. */
int btr_sa_gemm_nt_rc(int rows, int n, int k, const float *x0, const float *w0, const float *w,
                      int ldw, float *c, int ldc, const float *pa, const float *pb, float *part,
                      btr_stream_t stream);
int btr_sa_gemm_tn_rc(int rows, int n, int k, const float *g, int ldg, const float *x0,
                      const float *w0, const float *pa, const float *pb, float *pw, float *dw,
                      btr_stream_t stream);
int btr_sa_rc_wgrad_blocks(long long rows, int c);
int btr_sa_bn_relu_bwd_rc(long long rows, int c, int ldg, const float *g, const float *x0,
                          const float *w0, const float *scale, const float *shift,
                          const float *mean, const float *invstd, float *part, float *m1,
                          float *m2, float *dgamma, float *dbeta, float *pw, float *dw0,
                          btr_stream_t stream);

/* Pooled-layer backward without a dense dY pass.  btr_sa_pool_bwd_coef computes the BatchNorm
 * backward statistics of the max-pooled (last) layer like btr_sa_pool_bwd, but leaves y
 * untouched and returns dcl [b*m][c], alpha [c], beta [c] with
 *   dY[r][c] = alpha[c]*y[r][c] + beta[c] + (r % s == arg[r/s][c] ? dcl[r/s][c] : 0);
 * btr_sa_gemm_nt_pool (input gradient, dX = dY . W) and btr_sa_gemm_tn_pool (weight gradient,
 * dW = dY^T . f(x)) form dY inside their operand staging.  Replaces the F.max_pool2d / ReLU /
 * BatchNorm2d backward of pointnet2_modules.py:258-266 for the last SharedMLP layer. */
int btr_sa_pool_bwd_coef(int b, int m, int s, int c, int ldy, const float *y, const float *dout,
                         const float *out, const unsigned char *arg, const float *mean,
                         const float *invstd, const float *scale, const float *shift,
                         float *part, float *m1, float *m2, float *dgamma, float *dbeta,
                         float *dcl, float *alpha, float *beta, btr_stream_t stream);
/* Pooling epilogue: the last MLP layer's GEMM (prologue = previous layer's BN + ReLU, epilogue =
 * BN statistics) also emits, per group of `s` rows and column, the extremum of the pre-BN output
 * that the max-pool will select (the maximum where gamma >= 0, the minimum elsewhere: BatchNorm +
 * ReLU are monotone per channel and sign(scale) = sign(gamma)) and the row where it occurs first:
 * gext [rows/s][n] f32, aext [rows/s][n] u8.  btr_sa_pool_fin turns them into what btr_sa_pool
 * computes from the whole tensor.  btr_sa_gemm_nt_poolfwd_supported: n > 64, s in {16,32,64},
 * rows % s == 0. */
int btr_sa_gemm_nt_poolfwd_supported(int rows, int n, int s);
int btr_sa_gemm_nt_poolfwd(int rows, int n, int k, const float *a, int lda, const float *w,
                           int ldw, float *c, int ldc, const float *pa, const float *pb,
                           float *part, int s, const float *gamma, float *gext,
                           unsigned char *aext, btr_stream_t stream);
int btr_sa_pool_fin(int b, int m, int c, const float *gext, const unsigned char *aext,
                    const float *scale, const float *shift, float *out, float *out_cl,
                    unsigned char *arg, btr_stream_t stream);
/* Round 5: the pooled layer without its stored output.  btr_sa_gemm_nt_poolfwd accepts c == NULL
 * (statistics and extrema only) when btr_sa_gemm_nt_poolfwd_nostore_supported() returns 1 (the
 * streaming kernel's shapes; s = 8 for the block extrema of compact rows);  btr_sa_pool_fin_y /
 * btr_sac_pool_y also leave ywin (b, m, c) = the pre-BN value of the arg-max row, which
 * btr_sa_pool_bwd_coef takes in place of y with ldy == 0 (needs c % 4 == 0 and <= 1024 tiles of 64
 * centres); btr_sa_bwd_gram is the backward that goes with it. */
int btr_sa_gemm_nt_poolfwd_nostore_supported(int rows, int n, int k, int s);
int btr_sa_pool_fin_y(int b, int m, int c, const float *gext, const unsigned char *aext,
                      const float *scale, const float *shift, float *out, float *out_cl,
                      unsigned char *arg, float *ywin, btr_stream_t stream);
int btr_sac_pool_y(int b, int m, int c, const float *gext, const unsigned char *aext,
                   const int *goff, const float *scale, const float *shift, float *out,
                   float *out_cl, unsigned char *arg, float *ywin, btr_stream_t stream);
int btr_sa_gemm_nt_pool(int rows, int n, int k, const float *y, int ldy, const float *w, int ldw,
                        float *c, int ldc, int s, const unsigned char *arg, const float *dcl,
                        const float *alpha, const float *beta, btr_stream_t stream);
int btr_sa_gemm_tn_pool(int rows, int n, int k, const float *y, int ldy, int s,
                        const unsigned char *arg, const float *dcl, const float *alpha,
                        const float *beta, const float *x, int ldx, const float *pa,
                        const float *pb, float *pw, float *dw, btr_stream_t stream);

/* Single-launch INFERENCE set-abstraction layer (csrc/sa_mlp.hip sa_eval_fused_kernel): the eval
 * mode of PointnetSAModuleVotes.forward (pointnet2_modules.py:210-272 with module.eval(): BatchNorm
 * on its running statistics, given here as per-channel scale a_l / shift b_l) -- gather, three
 * 1x1 convolutions with BN + ReLU, max over the nsample axis -- without any rows x channels tensor
 * in HBM.  Covers the layer shape that carries the bytes: 3 * use_xyz + c <= 4 input columns,
 * widths c1, c2 <= 64 and c3 <= 128, nsample 16 / 32 / 64 (SA1: 4 -> 64 -> 64 -> 128, 64
 * samples); w0 [c1][4] (zero-padded input columns), w1 [c2][ld1], w2 [c3][ld2]; idx (b, m, s)
 * from btr_ball_query; out (b, c3, m) and its channel-last twin out_cl (b, m, c3). */
int btr_sa_eval_fused_supported(int s, int c, int use_xyz, int c1, int c2, int c3);
int btr_sa_eval_fused(int b, int n, int m, int s, int c, int use_xyz, float radius_div,
                      const float *xyz, const float *new_xyz, const float *feats_cl,
                      const int *idx, int c1, int c2, int c3, const float *w0, const float *w1,
                      int ld1, const float *w2, int ld2, const float *a0, const float *b0,
                      const float *a1, const float *b1, const float *a2, const float *b2,
                      float *out, float *out_cl, btr_stream_t stream);

/* A hidden layer's whole backward as ONE pass over its rows (csrc/sa_mlp.hip
 * sa_bwd_fused_kernel; round 4).  What the calls above do in five passes for layer l -- weight
 * gradient and input gradient each reading dY_l, then BatchNorm_{l-1}'s statistics pass and its
 * in-place apply pass over (dZ_{l-1}, Y_{l-1}) -- is one streaming kernel: dY_l is formed while
 * its rows are staged (arg != NULL: the pooled layer's gradient from y, arg, dcl, alpha, beta as
 * in btr_sa_gemm_*_pool with g = the pooled layer's pre-BN output; arg == NULL: BatchNorm_l's
 * backward dY = scale (m g - w (m1 + xhat m2)) from g = dZ_l, yl = Y_l and layer l's scale,
 * shift, mean, invstd, m1, m2), the same staged planes feed
 *   dw [n][k]    = dY_l^T . X_{l-1}   (pw: [btr_sa_bwd_fused_chunks()][n][k] partials)
 *   dz [rows][k] = dY_l . W_l         (wt = W_l^T [k][ldw]),
 * and the thread that stores a piece of dz also holds the matching raw Y_{l-1} values, so the
 * sums of BatchNorm_{l-1}'s backward (spart [chunks][2][k] -> m1, m2, dgamma, dbeta) come out of
 * the same pass.  X_{l-1} = relu(pa * y + pb), y = x [rows][ldx], or y rebuilt from the
 * 4-column input rows x [rows][4] and w0 [k][4] when w0 != NULL (first-layer recompute);
 * mu_p / is_p: mean and invstd of layer l-1.  The caller then feeds (dz, Y_{l-1}, m1, m2) to
 * the next call, or to btr_sa_bn_relu_bwd_rc_apply for a recomputed first layer.
 * Supported: n <= 256, k <= 512, multiples of 4, bf16x6 GEMMs on (BTR_BWD_FUSED=0: never).
 * btr_sa_bwd_fused_chunks(): the rows spart and the chunk count pw must be sized for.  (With
 * BTR_BWD_FUSED_SPLIT=1 -- measured slower, off -- n > 128 runs as two launches over the column
 * slabs [0, 128) and [128, n): the second adds its product onto the first one's dz and writes
 * its partial rows behind the first one's; the function then counts both.)
 * Reference: the autograd backward of SharedMLP's Conv2d + BatchNorm2d + ReLU stack,
 * pointnet2/pytorch_utils.py:11-36, 157-188. */
int btr_sa_bwd_fused_supported(int rows, int n, int k);
int btr_sa_bwd_fused_chunks(int rows, int n, int k);
int btr_sa_bwd_fused(int rows, int n, int k, const float *g, int ldg, const float *yl,
                     const float *sc, const float *sh, const float *mu, const float *is,
                     const float *m1l, const float *m2l, int s, const unsigned char *arg,
                     const float *dcl, const float *alpha, const float *beta, const float *x,
                     int ldx, const float *w0, const float *pa, const float *pb,
                     const float *mu_p, const float *is_p, const float *wt, int ldw, float *dz,
                     int ldz, float *pw, float *dw, float *spart, float *m1, float *m2,
                     float *dgamma, float *dbeta, btr_stream_t stream);
/* The POOLED layer's whole backward without its pre-BN output (csrc/sa_mlp.hip
 * sa_bwd_gram_kernel; round 5).  The dense part of the pooled layer's gradient is affine in
 * Y_l = X W_l^T -- dY[r][n] = w_r (alpha[n] Y_l[r][n] + beta[n]) + sparse[r][n], sparse = dcl on
 * the arg-max row of (group, channel), w_r = copies a compact row stands for -- so
 *   dz [rows][k] = w_r (x_r M + c) + sparse_r W_l,       M = W_l^T diag(alpha) W_l, c = beta^T W_l
 *   dw [n][k]    = diag(alpha) W_l G + beta (x) sx + sparse^T X,   G = sum_r w_r x_r x_r^T,
 *                                                                  sx = sum_r w_r x_r
 * need X = relu(pa * x + pb) (x [rows][ldx] = Y_{l-1}) only: per row k floats read and k written
 * instead of n + k read, the forward stores no Y_l (btr_sa_gemm_nt_poolfwd with c == NULL) and
 * the dense products are k x k instead of n x k per row.  Operands as btr_sa_bwd_fused with
 * arg != NULL, minus g; w = W_l [n][k] row-major (leading dimension k), wt = W_l^T [k][ldw];
 * pw [btr_sa_bwd_gram_chunks()][n][k], spart [chunks][2][k], gscratch
 * btr_sa_bwd_gram_scratch_floats() floats (256-byte aligned).  Outputs as btr_sa_bwd_fused.
 * Supported: btr_sa_bwd_fused's shapes with k <= 128 (BTR_POOL_GRAM=0: never).  Same result as
 * the Y_l-reading form up to float32 rounding (G and sx are combined in float64).
 * Reference: the autograd backward of max_pool2d over the last Conv2d + BatchNorm2d + ReLU of
 * SharedMLP, pointnet2_modules.py:262-267, pytorch_utils.py:11-36. */
int btr_sa_bwd_gram_supported(int rows, int n, int k);
int btr_sa_bwd_gram_chunks(int rows, int n, int k);
size_t btr_sa_bwd_gram_scratch_floats(int rows, int n, int k);
int btr_sa_bwd_gram(int rows, int n, int k, const float *x, int ldx, const float *pa,
                    const float *pb, const float *mu_p, const float *is_p, const float *w,
                    const float *wt, int ldw, int s, const unsigned char *arg, const float *dcl,
                    const float *alpha, const float *beta, float *dz, int ldz, float *pw,
                    float *dw, float *gscratch, float *spart, float *m1, float *m2,
                    float *dgamma, float *dbeta, btr_stream_t stream);
/* The two halves of btr_sa_bn_relu_bwd (statistics + finalisation; in-place apply): a caller
 * that hands (g, y, m1, m2) to btr_sa_bwd_fused only needs the first. */
int btr_sa_bn_relu_bwd_sums(long long rows, int c, int ld, const float *g, const float *y,
                            const float *scale, const float *shift, const float *mean,
                            const float *invstd, float *part, float *m1, float *m2,
                            float *dgamma, float *dbeta, btr_stream_t stream);
int btr_sa_bn_relu_bwd_apply(long long rows, int c, int ld, float *g, const float *y,
                             const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *m1, const float *m2,
                             btr_stream_t stream);
/* btr_sa_bn_relu_bwd_rc without its statistics pass: m1, m2 are given (btr_sa_bwd_fused). */
int btr_sa_bn_relu_bwd_rc_apply(long long rows, int c, int ldg, const float *g, const float *x0,
                                const float *w0, const float *scale, const float *shift,
                                const float *mean, const float *invstd, const float *m1,
                                const float *m2, float *pw, float *dw0, btr_stream_t stream);

/* ---- VoteNet loss, forward + backward (the caller right after the hot path).
 * Replaces the ~250 torch launches of detection/Votenet/models/loss_helper.py:336-400
 * (compute_vote_loss :24-69, compute_objectness_loss :111-152, compute_box_and_sem_cls_loss
 * :154-228; nn_distance utils/nn_distance.py:34-61) with three kernels.  vote_factor 1.
 *   net (b,cout,k): raw proposal-head output, channels as proposal_module.py:18-50
 *     [objectness 2 | centre offset 3 | heading scores nh | heading residuals nh |
 *      size scores ns | size residuals ns*3 | semantic nc];  agg_xyz (b,k,3); vote_xyz,
 *   seed_xyz (b,s1,3); seed_inds (b,s1) i32 into the n input points; labels as the dataset
 *   batch dict (scannet_detection_dataset.py:197-219): vote_label (b,n,9), vote_label_mask
 *   (b,n) i64, center_label (b,k2,3), box_label_mask (b,k2), heading_class_label (b,k2) i64,
 *   heading_residual_label (b,k2), size_class_label (b,k2) i64, size_residual_label (b,k2,3),
 *   sem_cls_label (b,k2) i64, mean_size (ns,3).
 * Outputs: objectness_label (b,k) i64, objectness_mask (b,k) f32, object_assignment (b,k) i64;
 *   stats[14] = loss, vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg,
 *   sem_cls, box, pos_ratio, neg_ratio, obj_acc, loss once more (a word a caller may hand out as
 *   the loss tensor and edit in place without touching the reported entries).  Kept for the backward: j1c (b,k) i32,
 *   k2c (b,k2) i32, vote_arg (b,s1) i8, part (b,16) f32, norm (4) f32.
 * weights8 (HOST pointer): weights of (vote, objectness, center, heading_cls, heading_reg,
 *   size_cls, size_reg, sem_cls) in loss/10 -- get_loss: {1, .5, 1, .1, 1, .1, 1, .1}.
 * vote_mode 1 + i2v (b,k2) i32 scratch: the weakly supervised vote term of the
 *   Back-to-Reality loss (compute_weak_vote_loss, loss_helper.py:71-109; get_loss_DA
 *   :548-664 is two such calls with the source / target weights) instead of
 *   compute_vote_loss. */
int btr_votenet_loss_fwd(int b, int k, int k2, int nh, int ns, int nc, int s1, int n, int cout,
                         const float *net, const float *agg_xyz, const float *vote_xyz,
                         const float *seed_xyz, const int *seed_inds, const float *vote_label,
                         const long long *vote_label_mask, const float *center_label,
                         const float *box_label_mask, const long long *heading_class_label,
                         const float *heading_residual_label, const long long *size_class_label,
                         const float *size_residual_label, const long long *sem_cls_label,
                         const float *mean_size, long long *objectness_label,
                         float *objectness_mask, long long *object_assignment, int *j1c, int *k2c,
                         signed char *vote_arg, float *part, float *stats, float *norm,
                         const float *weights8, int vote_mode, int *i2v, btr_stream_t stream);
/* dnet (b,cout,k), dagg (b,k,3), dvote (b,s1,3) <- d loss / d input, times gout[0]. */
int btr_votenet_loss_bwd(int b, int k, int k2, int nh, int ns, int nc, int s1, int n, int cout,
                         const float *gout, const float *norm, const float *net,
                         const float *agg_xyz, const float *vote_xyz, const float *seed_xyz,
                         const int *seed_inds, const float *vote_label,
                         const long long *vote_label_mask, const float *center_label,
                         const float *box_label_mask, const long long *heading_class_label,
                         const float *heading_residual_label, const long long *size_class_label,
                         const float *size_residual_label, const long long *sem_cls_label,
                         const float *mean_size, const long long *objectness_label,
                         const float *objectness_mask, const long long *object_assignment,
                         const int *j1c, const int *k2c, const signed char *vote_arg, float *dnet,
                         float *dagg, float *dvote, const float *weights8, int vote_mode,
                         const int *i2v, btr_stream_t stream);

/* The domain-adaptation term of get_loss_DA (reference: detection/Votenet/models/loss_helper.py:
 * 618-650 with FocalLoss, :466-545; GroupFree3D's loss_helper.py:673-712): per branch coef *
 * mean(e(l)^2 * objectness_label) + coef * focal(softmax(global_d_pred), domain, gamma), e(l) = l
 * (source, domain 0) / 1 - l (target, domain 1); coef 0.5 for VoteNet, 1 for GroupFree3D.  global_* (b, 2) logits, local_* (b, k) sigmoid outputs, label_* (b, k) i64.
 * out[3] = (total, source part, target part); grads (4 b + 2 b k floats) = the gradient for a unit
 * upstream gradient, [d global_S | d local_S | d global_T | d local_T].  One launch. */
int btr_domain_loss(int b, int k, float gamma, float coef, const float *global_S, const float *local_S,
                    const long long *label_S, const float *global_T, const float *local_T,
                    const long long *label_T, float *out, float *grads, btr_stream_t stream);

/* out (b,m,c) <- src (b,n,c)[idx (b,m)]: row gather of a channel-last tensor.  No counterpart of
 * its own in the reference: it does transpose + gather_points + transpose for `new_xyz`
 * (pointnet2/pointnet2_modules.py:238-240); same values. */
int btr_gather_rows(int b, int n, int m, int c, const float *src, const int *idx, float *out,
                    btr_stream_t stream);

/* Its gradient: grad_src (b,n,c) = 0, then grad_src[idx (b,m)] += grad_out (b,m,c) -- what the
 * reference's route computes with gather_points_grad (sampling.cpp:46-69) between two transposes;
 * one launch, grad_src need not be cleared by the caller.  Repeated indices add up. */
int btr_gather_rows_grad(int b, int n, int m, int c, const float *grad_out, const int *idx,
                         float *grad_src, btr_stream_t stream);

/* ---- Evaluation-side box arithmetic (the caller after the forward at eval time).
 * The reference runs these per box in numpy / scipy on the host, in float64; so do the kernels.
 *
 * btr_nms_boxes: greedy NMS, one scene per workgroup -- nms_2d_faster / nms_3d_faster /
 *   nms_3d_faster_samecls (detection/Votenet/utils/nms.py:42-156).  boxes (b,k,2*dim) f64 =
 *   [min_0..min_{dim-1}, max_0..max_{dim-1}], dim 2 or 3; score (b,k) f64; cls (b,k) i32 or
 *   NULL (NULL = boxes of different classes suppress each other too); valid (b,k) u8 or NULL
 *   (rows the reference filters out before the call, ap_helper.py:145,163,183); old_type: the
 *   overlap is inter / area(candidate) instead of the IoU.  pick (b,k) u8 <- 1 for kept boxes.
 *   k <= 1024.  Equal scores: the higher index is visited first (numpy leaves it unspecified).
 * btr_points_in_boxes: count (b,k) i32 <- min(cap, number of points of the scene inside box j)
 *   -- `remove_empty_box` (models/ap_helper.py:116-127, extract_pc_in_box3d :27-30) without the
 *   Delaunay triangulation per box.  points (b,n,point_stride) f32 in upright-depth coordinates
 *   (the network input); center (b,k,3), size (b,k,3) = (l,w,h), angle (b,k): the box in
 *   upright-camera coordinates as get_3d_box takes it (utils/box_util.py:211-227).
 * btr_box3d_iou: iou (s,p,g) f64 <- box3d_iou (utils/box_util.py:98-128) of corners1 (s,p,8,3)
 *   with corners2 (s,g,8,3), corner order of get_3d_box, upright-camera coordinates. */
int btr_nms_boxes(int b, int k, int dim, const double *boxes, const double *score,
                  const int *cls, const unsigned char *valid, double threshold, int old_type,
                  unsigned char *pick, btr_stream_t stream);
int btr_points_in_boxes(int b, int n, int k, int point_stride, int cap, const float *points,
                        const double *center, const double *size, const double *angle,
                        int *count, btr_stream_t stream);
int btr_box3d_iou(int nscene, int p, int g, const double *corners1, const double *corners2,
                  double *iou, btr_stream_t stream);

/* ---- ball query over the FPS's spatial sort ---------------------------------------------------
 * Same result as btr_ball_query (ball_query_gpu.cu:14-49).  `fps_workspace`: the workspace a
 * preceding btr_furthest_point_sampling_ws call ON THE SAME xyz (n > 4096) was given; it still
 * holds the cloud sorted along a Hilbert curve in 64-point buckets, which this call searches
 * instead of counting-sorting the cloud again (csrc/ball_query_bucket.hip).  The workspace
 * size is 0 when the shape is not supported (use btr_ball_query_ws then).
 * CONTRACT: the WHOLE fps_workspace must stay untouched between the two calls, on the same host
 * thread: besides the sorted points the FPS kernel leaves its per-bucket bounding boxes in the
 * part of the workspace its counting sort no longer needs, and this call uses them when the
 * calling thread's last FPS launch on that address had the same (b, n).  Each box carries the
 * launch's epoch and its own position; every query workgroup checks all stamps of its scene and,
 * on any mismatch (area overwritten / partially restored / filled by another launch), bounds
 * the buckets from the sorted points instead -- slower, same result.  The sorted POINTS carry no
 * such check: handing this call a workspace whose point area is not the FPS's sort of `xyz`
 * gives wrong neighbour lists, exactly like handing btr_ball_query the wrong xyz. */
size_t btr_ball_query_buckets_workspace_bytes(int b, int n, int m, int nsample);
int btr_ball_query_buckets(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                           const void *fps_workspace, int *idx, void *workspace,
                           size_t workspace_bytes, btr_stream_t stream);

/* ---- compact rows of the fused set-abstraction path ------------------------------------------
 * The reference runs its shared MLP over ALL nsample rows of every group although a ball-query
 * row is its distinct hits followed by copies of the first one (ball_query_gpu.cu:39-43;
 * grouping: pointnet2_utils.py:346-357; MLP: pointnet2_modules.py:243-252).  A copy computes
 * what its original computes, so the btr_sa_* kernels can evaluate the layer on the distinct
 * rows only -- every group keeps ceil8(#distinct) rows, its first row carries the weight of
 * the copies it stands for in the BatchNorm statistics (sa_mlp.hip, "compact rows").
 * btr_sac_plan builds the description on the device (the row count is data dependent and
 * never read by the host); btr_sac_bind(&cm) makes every following btr_sa_gemm_* /
 * btr_sa_bn_relu_bwd* / btr_sa_pool_bwd_coef call ON THIS HOST THREAD operate on compact rows
 * (their `rows` argument then only sizes grids: pass the dense row count), btr_sac_bind(NULL)
 * ends it.  max_rows = groups * nsample (dense worst case) sizes every buffer. */
typedef struct btr_compact {
  const int *dims;     /* [2] device: rows, rows / 8                                         */
  const float *bw;     /* [max_rows / 8] device: weight of each 8-row block's first row      */
  const int *bgrp;     /* [max_rows / 8] device: group of each block                         */
  const int *goff;     /* [groups + 1] device: first row of each group                       */
  double dense_rows;   /* groups * nsample: BatchNorm's element count                        */
} btr_compact_t;
void btr_sac_bind(const btr_compact_t *cm);
int btr_sac_plan(int groups, int s, const int *idx, int *len_tmp /*[groups]*/, int *goff,
                 int *dims, int *cidx /*[max_rows]*/, int *bgrp, float *bw, btr_stream_t stream);
int btr_sac_gather(int b, int n, int m, int max_rows, int c, int ldx, int use_xyz,
                   float radius_div, const float *xyz, const float *new_xyz,
                   const float *feats_cl, const int *cidx, const int *bgrp, const int *dims,
                   float *x0, btr_stream_t stream);
/* max-pool from the per-8-row-block extrema of btr_sa_gemm_nt_poolfwd(..., s = 8, ...) */
int btr_sac_pool(int b, int m, int c, const float *gext, const unsigned char *aext,
                 const int *goff, const float *scale, const float *shift, float *out,
                 float *out_cl, unsigned char *arg, btr_stream_t stream);
size_t btr_sac_scatter_workspace_bytes(int b, int n, int max_rows);
int btr_sac_scatter(int b, int n, int m, int c, int ldx, int use_xyz, const float *dx0,
                    const int *cidx, const int *goff, float *dfeat_cl, void *workspace,
                    size_t workspace_bytes, int max_rows, btr_stream_t stream);

/* ---- point-wise MLP chains (feature propagation, vote generator, proposal head) --------------
 * reference: pytorch_utils.SharedMLP / Conv1d + BatchNorm1d + ReLU sequences
 * (pointnet2_modules.py:469-514, models/voting_module.py:37-56, models/proposal_module.py:75-113)
 * on channel-last rows (B*N, C).  btr_pm_gemm_nt = btr_sa_gemm_nt on 64-row tiles (these
 * problems have 2 048 - 16 384 rows) with an optional bias; the BatchNorm statistics /
 * backward / weight-gradient entry points of the btr_sa_* family are shared. */
int btr_pm_gemm_grid(int rows);
int btr_pm_gemm_nt(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                   float *c, int ldc, const float *pa, const float *pb, float *part,
                   const float *bias, btr_stream_t stream);
/* Round 5: the small-M form (csrc/sa_mlp.hip gemm_nt_sm_kernel; <= 16 384 rows, k <= 576): 32 x 64
 * tiles, the whole k extent in at most two staged chunks, and the weights as three bf16 planes
 * [3][n][ceil16(k)] (btr_pm_weight_planes; btr_pm_weight_planes_bytes() bytes) that every row
 * tile loads as MFMA fragments instead of splitting W again.  Operands otherwise as btr_pm_gemm_nt;
 * same bf16x6 products, the two k halves added in another order.  BTR_PM_SM=0: never supported. */
size_t btr_pm_weight_planes_bytes(int n, int k);
int btr_pm_weight_planes(int n, int k, const float *w, int ldw, void *planes, btr_stream_t stream);
int btr_pm_gemm_nt_sm_supported(int rows, int n, int k);
int btr_pm_gemm_nt_sm(int rows, int n, int k, const float *a, int lda, const void *planes,
                      float *c, int ldc, const float *pa, const float *pb, float *part,
                      const float *bias, btr_stream_t stream);
int btr_pm_out(int b, int n, int c, int ldy, const float *y, const float *scale,
               const float *shift, int relu, float *out_bcn, float *out_cl, btr_stream_t stream);
int btr_pm_rows(int b, int n, int c, int ldr, const float *x, float *rows, btr_stream_t stream);

/* ---- whole-layer entry points: one call per set-abstraction layer / per MLP chain ------------
 * The btr_sa_* / btr_pm_* launches above, sequenced in C++ (csrc/sa_layer.hip) instead of by
 * the caller: a training step is ~400 kernel launches, and a host that issues them one binding
 * call at a time needs longer than the GPU needs to run them.  One description (btr_*_t, filled
 * by the caller), one plan (btr_*_plan_t: which variant runs + where every tensor lives inside
 * ONE saved buffer, ONE scratch buffer and ONE flat gradient buffer), one call forward, one
 * call backward.  Results are bit-identical to issuing the individual calls.
 *
 * reference: PointnetSAModuleVotes.forward (pointnet2_modules.py:243-267) = QueryAndGroup
 * (pointnet2_utils.py:317-376) + SharedMLP (pytorch_utils.py:11-36) + max_pool2d, and its
 * autograd backward; SharedMLP / Conv1d+BatchNorm1d+ReLU chains for btr_pm_chain_*. */
#define BTR_MAX_LAYERS 8
enum {
  BTR_SA_OPT_COMPACT = 1,       /* distinct-neighbour rows for nsample >= 32                     */
  BTR_SA_OPT_RECOMPUTE = 2,     /* never store the first layer's output of a 4-column input      */
  BTR_SA_OPT_POOL_EPILOGUE = 4, /* group extrema from the last GEMM's epilogue                    */
  BTR_SA_OPT_POOL_GRAD = 8,     /* pooled gradient formed inside the GEMM operand staging        */
  BTR_SA_OPT_POOL_GRAM = 16,    /* pooled layer without its stored output (btr_sa_bwd_gram); the
                                   plan reports it as pool_grad == 2                              */
  BTR_SA_OPT_PPFL = 64,         /* per-point first layer: W_f f_j once per point instead of once
                                   per (centre, neighbour) row, the gradients' feature parts as
                                   products over the points (levels with >= 32 feature channels
                                   and no coordinate gradient; csrc/sa_mlp.hip
                                   ppfl_gather_add_kernel)                                        */
  BTR_SA_OPT_PPFL_XYZ = 128,    /* ... also where coordinate gradients are asked for (the vote
                                   aggregation; dense rows): d rel = dY_0 W_x as one more 4-column
                                   product.  Measured neutral, so the Python layer leaves it off
                                   unless BTR_SA_PPFL_XYZ=1                                       */
  BTR_SA_OPT_EVAL = 256         /* inference (module.eval() under no_grad, the evaluation pass of
                                   train_Votenet_FSB.py:246-293): gamma[l] / beta[l] ARE the
                                   affine map y -> a y + b of BatchNorm on its running statistics
                                   (a = gamma / sqrt(running_var + eps), b = beta - running_mean
                                   a, derived by the caller), running_mean / running_var /
                                   num_batches_tracked NULL; the forward is the training forward's
                                   kernels (compact rows, per-point first layer, streaming GEMMs,
                                   pool in the epilogue) minus the finalisers; no backward        */
};
typedef struct {
  int b, n, m, s, c;            /* batch, points, centres, nsample, feature channels (may be 0)  */
  int use_xyz;
  float radius_div;             /* radius when normalize_xyz, else 1                             */
  int layers;
  int width[BTR_MAX_LAYERS];    /* output channels per layer (multiples of 4)                    */
  const float *w[BTR_MAX_LAYERS];     /* (width[l], in_l) row-major, in_0 = 3*use_xyz + c        */
  const float *gamma[BTR_MAX_LAYERS];
  const float *beta[BTR_MAX_LAYERS];
  float *running_mean[BTR_MAX_LAYERS];  /* NULL: not tracked                                     */
  float *running_var[BTR_MAX_LAYERS];
  long long *num_batches_tracked[BTR_MAX_LAYERS];  /* NULL or device int64, += 1                 */
  float eps[BTR_MAX_LAYERS];
  float momentum[BTR_MAX_LAYERS];
  int need_dxyz, need_dnew_xyz, need_dfeat;  /* which input gradients the backward must produce  */
  int options;                  /* BTR_SA_OPT_* the caller allows                                */
} btr_sa_layer_t;

typedef struct {
  int compact, recompute, pool_epilogue, pool_grad;  /* what will run                            */
  int rows, k0, k0p;            /* b*m*s, 3*use_xyz + c, the same rounded up to 4                */
  int kin[BTR_MAX_LAYERS];      /* padded input width per layer                                  */
  /* byte offsets into `saved` */
  size_t x0, y[BTR_MAX_LAYERS], w2[BTR_MAX_LAYERS], wt[BTR_MAX_LAYERS], stats[BTR_MAX_LAYERS];
  size_t arg, goff, dims, cidx, bgrp, bw;
  size_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes;
  /* float offsets into `grads`: dW (width[l], kin[l]) padded, dgamma, dbeta.  Layer 0 without the
   * first-layer recompute: dW_0 is written as DENSE (width[0], 3 * use_xyz + c) rows at dw[0] --
   * the parameter's own shape, without the columns the 4-aligned kin[0] added (the block keeps its
   * padded size) */
  size_t dw[BTR_MAX_LAYERS], dgamma[BTR_MAX_LAYERS], dbeta[BTR_MAX_LAYERS];
  size_t grads_floats;
} btr_sa_plan_t;

int btr_sa_layer_plan(const btr_sa_layer_t *d, btr_sa_plan_t *plan);
/* 1 when this (description, plan) runs its first layer per point (BTR_SA_OPT_PPFL, covered shape) */
int btr_sa_layer_ppfl(const btr_sa_layer_t *d, const btr_sa_plan_t *plan);
/* out (b, width[L-1], m); out_cl (b, m, width[L-1]); feats_cl (b, n, c) channel-last or NULL */
int btr_sa_layer_forward(const btr_sa_layer_t *d, const btr_sa_plan_t *plan, const float *xyz,
                         const float *new_xyz, const float *feats_cl, const int *idx, float *out,
                         float *out_cl, void *saved, void *scratch, btr_stream_t stream);
/* dfeat (b, c, n), dxyz (b, n, 3), dnew_xyz (b, m, 3): NULL when not needed.  `saved` is
 * consumed (the pooled layer's output may be overwritten): one backward per forward. */
int btr_sa_layer_backward(const btr_sa_layer_t *d, const btr_sa_plan_t *plan, const int *idx,
                          const float *out, const float *dout, void *saved, float *grads,
                          float *dfeat, float *dxyz, float *dnew_xyz, void *scratch,
                          btr_stream_t stream);

typedef struct {
  int b, n, c;                  /* x: (b, c, n); rows are zero-padded to a multiple of 4 columns */
  int layers;
  int width[BTR_MAX_LAYERS];
  int has_bn[BTR_MAX_LAYERS];   /* conv -> BatchNorm -> ReLU; the last layer may be a bare conv  */
  const float *w[BTR_MAX_LAYERS];     /* (width[l], in_l) row-major                              */
  const float *bias[BTR_MAX_LAYERS];  /* NULL or (width[l]); skipped in front of a BatchNorm     */
  const float *gamma[BTR_MAX_LAYERS];
  const float *beta[BTR_MAX_LAYERS];
  float *running_mean[BTR_MAX_LAYERS];
  float *running_var[BTR_MAX_LAYERS];
  long long *num_batches_tracked[BTR_MAX_LAYERS];
  float eps[BTR_MAX_LAYERS];
  float momentum[BTR_MAX_LAYERS];
  int need_dx;
} btr_pm_chain_t;

typedef struct {
  int rows;
  int np[BTR_MAX_LAYERS];       /* width rounded up to 4                                         */
  int kin[BTR_MAX_LAYERS];
  size_t x0, y[BTR_MAX_LAYERS], w2[BTR_MAX_LAYERS], wt[BTR_MAX_LAYERS], stats[BTR_MAX_LAYERS];
  size_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes;
  size_t dw[BTR_MAX_LAYERS], dbias[BTR_MAX_LAYERS], dgamma[BTR_MAX_LAYERS], dbeta[BTR_MAX_LAYERS];
  size_t grads_floats;
} btr_pm_plan_t;

int btr_pm_chain_plan(const btr_pm_chain_t *d, btr_pm_plan_t *plan);
/* x_bcn (b, c, n) or, when the producer kept it, x_cl (b*n, c) (then x_bcn may be NULL);
 * out (b, width[L-1], n), out_cl (b*n, width[L-1]); out may be NULL when the last layer is a bare
 * convolution of a width that is a multiple of 4 (its GEMM then writes out_cl itself) */
int btr_pm_chain_forward(const btr_pm_chain_t *d, const btr_pm_plan_t *plan, const float *x_bcn,
                         const float *x_cl, float *out, float *out_cl, void *saved,
                         void *scratch, btr_stream_t stream);
/* x_cl: the rows the forward was given (NULL when it was given x_bcn: they are in `saved`) */
int btr_pm_chain_backward(const btr_pm_chain_t *d, const btr_pm_plan_t *plan, const float *x_cl,
                          const float *dout, void *saved, float *grads, float *dx,
                          void *scratch, btr_stream_t stream);

/* ---- whole-backbone entry points (csrc/backbone.hip) ------------------------------------------
 * reference: Pointnet2Backbone.forward (models/backbone_module.py:83-133): L set-abstraction
 * levels (PointnetSAModuleVotes, pointnet2_modules.py:210-272) and F < L feature-propagation
 * modules (PointnetFPModule, :469-514), and the autograd backward of that graph.  Module j
 * (0-based) interpolates the features living on level L - j (SA_L's output for j = 0, module
 * j-1's output otherwise) onto the points of level L - j - 1 and concatenates that level's SA
 * output (the skip connection) in front of its SharedMLP.
 * Five caller-provided arenas, laid out by btr_backbone_plan (byte offsets in the plan):
 *   geom     what depends on coordinates only: per level the FPS indices (b, m) i32, the
 *            sampled coordinates (b, m, 3), the ball-query lists (b, m, s) i32; per module
 *            the 3-NN indices / blend weights (b, n, 3); the (b, n, 3) / (b, n, c) split of the
 *            cloud when it carries features; scratch of the index kernels
 *   out      per level and module the output as (b, c, m) and channel-last (b, m, c)
 *   saved    what the backward reads (per layer as btr_sa_layer_* / btr_pm_chain_* lay it out)
 *   scratch  forward / backward scratch (plan->fwd_scratch_bytes / bwd_scratch_bytes)
 *   grads    ONE flat f32 buffer: the layers' gradient blocks back to back (plan->gr_sa / gr_fp
 *            give each layer's first float; inside a block the layer plan's dw/dgamma/dbeta)
 * btr_backbone_sampling fills `geom` from the cloud (b, n, 3 + c).  It may run on any stream,
 * e.g. a whole step ahead of the forward that consumes it (the caller orders the two).
 * `side` != NULL and != stream: level 1 runs on `stream`, the deeper levels and the 3-NN weights
 * on `side` (forked behind level 1), and btr_backbone_forward(..., wait_side = 1, ...) CALLED
 * NEXT ON THE SAME HOST THREAD waits for each of them right before the layer that needs it --
 * SA1's MLP then overlaps the remaining FPS levels.
 * btr_backbone_backward: dout_sa[l] (b, c_l, m_l) / dout_fp[j] (b, c, n) are the gradients of
 * the corresponding outputs, NULL entries (or NULL arrays) = no gradient. */
#define BTR_MAX_LEVELS 4
typedef struct {
  int b, n, c;                         /* the cloud: (b, n, 3 + c) f32                          */
  int levels, fps;                     /* L set-abstraction levels, F < L propagation modules   */
  float radius[BTR_MAX_LEVELS];        /* ball-query radius per level                           */
  btr_sa_layer_t sa[BTR_MAX_LEVELS];   /* sa[l].n / .c continue level l's m / last width        */
  btr_pm_chain_t fp[BTR_MAX_LEVELS];   /* fp[j].n = points of level L-j-1, .c = c_known + c_skip */
} btr_backbone_t;

typedef struct {
  btr_sa_plan_t sa[BTR_MAX_LEVELS];
  btr_pm_plan_t fp[BTR_MAX_LEVELS];
  /* geometry arena */
  size_t g_xyz, g_feat;
  size_t g_inds[BTR_MAX_LEVELS], g_new_xyz[BTR_MAX_LEVELS], g_idx[BTR_MAX_LEVELS];
  size_t g_fps_ws[BTR_MAX_LEVELS], g_fps_ws_bytes[BTR_MAX_LEVELS], g_fps_temp[BTR_MAX_LEVELS];
  int bq_buckets[BTR_MAX_LEVELS];      /* ball query over the FPS's spatial sort                */
  size_t g_nn_idx[BTR_MAX_LEVELS], g_nn_w[BTR_MAX_LEVELS];
  /* prepared with the sampling for forward / backward: the compact-row plans (compact levels)
   * and the filled inverted neighbour lists of the input-gradient scatters / of the 3-NN
   * gradient (0 bytes: not needed) */
  size_t g_goff[BTR_MAX_LEVELS], g_dims[BTR_MAX_LEVELS], g_cidx[BTR_MAX_LEVELS],
      g_bgrp[BTR_MAX_LEVELS], g_bw[BTR_MAX_LEVELS], g_len[BTR_MAX_LEVELS];
  size_t g_scat[BTR_MAX_LEVELS], g_scat_bytes[BTR_MAX_LEVELS];
  size_t g_ti[BTR_MAX_LEVELS], g_ti_bytes[BTR_MAX_LEVELS];
  size_t g_ws, g_ws_bytes, geom_bytes;
  /* output arena */
  size_t o_sa[BTR_MAX_LEVELS], o_sa_cl[BTR_MAX_LEVELS], o_fp[BTR_MAX_LEVELS],
      o_fp_cl[BTR_MAX_LEVELS], out_bytes;
  /* saved arena: per layer its own `saved` block; s_fpx = the modules' input rows */
  size_t s_sa[BTR_MAX_LEVELS], s_fp[BTR_MAX_LEVELS], s_fpx[BTR_MAX_LEVELS], saved_bytes;
  size_t fwd_scratch_bytes, bwd_scratch_bytes;
  size_t gr_sa[BTR_MAX_LEVELS], gr_fp[BTR_MAX_LEVELS], grads_floats;
} btr_backbone_plan_t;

int btr_backbone_plan(const btr_backbone_t *d, btr_backbone_plan_t *plan);
int btr_backbone_sampling(const btr_backbone_t *d, const btr_backbone_plan_t *plan,
                          const float *cloud, void *geom, btr_stream_t stream,
                          btr_stream_t side);
int btr_backbone_forward(const btr_backbone_t *d, const btr_backbone_plan_t *plan,
                         const float *cloud, const void *geom, void *out, void *saved,
                         void *scratch, int wait_side, btr_stream_t stream);
/* Software pipelining hook: `event` (a hipEvent_t) is recorded by the NEXT btr_backbone_forward
 * of the calling host thread, on its stream, right behind set-abstraction level `level`
 * (1-based; at the last level at the latest).  A trainer that computes the next batch's sampling
 * pyramid on a side stream makes that stream wait for the event: the ~2 000-step FPS chain then
 * runs beside the latency-bound middle of the step (SA3 .. loss .. their backward) instead of
 * beside the two bandwidth-bound levels (votenet/train.py train_step). */
void btr_backbone_fork_event(void *event, int level);
int btr_backbone_backward(const btr_backbone_t *d, const btr_backbone_plan_t *plan,
                          const void *geom, const void *out, const float *const *dout_sa,
                          const float *const *dout_fp, void *saved, float *grads, void *scratch,
                          btr_stream_t stream);

/* ---- vote assembly (models/voting_module.py:57-64, vote_factor 1) -----------------------------
 * from the generator's last layer on channel-last rows net_cl (b*n, ld_net >= 3 + c):
 * vote_xyz (b, n, 3) = seed_xyz + net[..., 0:3]; vote features = seed features + net[..., 3:],
 * written as (b, c, n) and channel-last (b*n, c).  The backward assembles the gradient w.r.t. the
 * generator's (b, 3 + c, n) output; the seed features' gradient is d vote_feat itself.
 * nrm != NULL (b*n floats, c <= 256): the features are also L2-normalised over the channels, what
 * VoteNet.forward does next (models/votenet.py:98-99); the backward then needs the normalised
 * features (channel-last) and nrm back and also writes the seed features' gradient. */
int btr_vote_assemble(int b, int n, int c, const float *net_cl, int ld_net,
                      const float *seed_xyz, const float *seed_cl, float *vote_xyz,
                      float *vote_feat_bcn, float *vote_feat_cl, float *nrm,
                      btr_stream_t stream);
int btr_vote_assemble_bwd(int b, int n, int c, const float *dvote_xyz, const float *dvote_feat_bcn,
                          const float *vote_feat_cl, const float *nrm, float *dnet_bcn,
                          float *dseed_bcn, btr_stream_t stream);

/* ---- fused multi-head attention core (GroupFree3D decoder) -----------------------
 * reference: detection/GroupFree3D/models/transformer.py:36-76 -> models/multi_head_attention.py
 * (softmax(q k^T / sqrt(d)) -> dropout -> . v per head, and its autograd backward).
 * q[l][b][h*d + c] is read at q + l*q_sl + b*q_sb (element strides), k / v likewise with
 * kv_sl / kv_sb: the projection outputs are used in place.  out, dout: (lq, b, h*d) contiguous;
 * lse, dsum: (b*h, lq) floats (saved log-sum-exp / backward scratch).  dq / dk / dv are written
 * with their own strides (e.g. into one packed (L, B, 3E) gradient).  dropout_p in [0, 1):
 * keep-mask = hash(seed, *step, element), identical in forward and backward for equal
 * (seed, *step); step may be NULL.  Head width d <= 64. */
int btr_attention_supported(int d);
int btr_attention_fwd(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                      long long q_sb, const float *k, const float *v, long long kv_sl,
                      long long kv_sb, float *out, float *lse, float scale, float dropout_p,
                      unsigned long long seed, const long long *step, btr_stream_t stream);
int btr_attention_bwd(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                      long long q_sb, const float *k, const float *v, long long kv_sl,
                      long long kv_sb, const float *out, const float *dout, const float *lse,
                      float *dsum, float *dq, long long dq_sl, long long dq_sb, float *dk,
                      float *dv, long long dkv_sl, long long dkv_sb, float scale, float dropout_p,
                      unsigned long long seed, const long long *step, btr_stream_t stream);

/* ---- whole decoder layer (csrc/decoder.hip) ---------------------------------------
 * reference: detection/GroupFree3D/models/transformer.py:11-76 (TransformerDecoderLayer.forward:
 * self-attention over the query points, cross-attention onto the seed points, FFN, post-norm;
 * position embeddings added to queries, keys and values) and its autograd backward, one call
 * each.  Parameters in torch's layouts (nn.MultiheadAttention in_proj_weight (3E, E) /
 * in_proj_bias / out_proj, nn.Linear, nn.LayerNorm).  Rows are batch-major channel-last:
 * x_cl / qpos_cl / out_cl (b*pq, e), key_cl / kpos_cl (b*pk, e) -- the layout the point-wise MLP
 * chains keep as their second output; qpos_cl / kpos_cl may be NULL (no position embedding).
 * Tensors at the module boundary are (b, e, p) like the reference's.  dropout in [0, 1): the
 * four nn.Dropout of the layer and the two attention dropouts draw their keep-masks from
 * hash(seed, *step, element); backward must see the (seed, *step) of its forward. */
typedef struct {
  int b, pq, pk, e, heads, ff;
  float dropout;
  unsigned long long seed;
  const long long *step;          /* device pointer or NULL                                      */
  const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;   /* self_attn                          */
  const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;   /* multihead_attn                     */
  const float *lin1_w, *lin1_b, *lin2_w, *lin2_b;         /* linear1 (ff, e), linear2 (e, ff)   */
  const float *ln_w[3], *ln_b[3];                         /* norm1, norm2, norm3                */
  float ln_eps[3];
} btr_decoder_layer_t;

typedef struct {
  int rq, rk;                     /* b*pq, b*pk                                                  */
  /* byte offsets into `saved` */
  size_t qp0, qkv, a1, lse1, xh1, rs1, x1, qp1, q2, kp, kv, a2, lse2, xh2, rs2, x2, h, xh3, rs3;
  size_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes;
  /* float offsets into `grads`; g_ln[i]: weight gradient then bias gradient (2e floats) */
  size_t g_sa_in_w, g_sa_in_b, g_sa_out_w, g_sa_out_b, g_ca_in_w, g_ca_in_b, g_ca_out_w,
      g_ca_out_b, g_lin1_w, g_lin1_b, g_lin2_w, g_lin2_b, g_ln[3];
  size_t grads_floats;
} btr_decoder_plan_t;

int btr_decoder_layer_plan(const btr_decoder_layer_t *d, btr_decoder_plan_t *plan);
/* out_cl (b*pq, e) and, when non-NULL, out_bcp (b, e, pq) */
int btr_decoder_layer_forward(const btr_decoder_layer_t *d, const btr_decoder_plan_t *plan,
                              const float *x_cl, const float *key_cl, const float *qpos_cl,
                              const float *kpos_cl, float *out_bcp, float *out_cl, void *saved,
                              void *scratch, btr_stream_t stream);
/* dout_bcp (b, e, pq); grads: plan->grads_floats floats, fully written; dx_bcp (b, e, pq),
 * dqpos_bcp (b, e, pq), dkey_bcp (b, e, pk) [= the gradient of kpos as well]: each may be NULL */
int btr_decoder_layer_backward(const btr_decoder_layer_t *d, const btr_decoder_plan_t *plan,
                               const float *x_cl, const float *key_cl, const float *qpos_cl,
                               const float *kpos_cl, const float *dout_bcp, void *saved,
                               float *grads, float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                               void *scratch, btr_stream_t stream);

/* ---- GroupFree3D: per-head loss and its gradient (csrc/gf_loss.hip) --------------
 * reference: detection/GroupFree3D/models/loss_helper.py:81-275
 * (compute_objectness_loss_based_on_query_points + compute_box_and_sem_cls_loss, smooth-L1 forms)
 * over all `heads` prediction heads at once.  heads[h]: the raw output of PredictHead h, (b, c, p)
 * with c = 4 + 2 nh + 4 ns + nc channels in the reference's order (objectness, centre residual,
 * heading scores, heading residuals, size scores, size residuals, semantic scores);
 * base_xyz (b, p, 3); seed_inds (b, s1) / sample_inds (b, p) int32; labels as the GroupFree3D
 * loader provides them (i64 class labels, f32 residuals); mean_size (ns, 3).
 * w_obj / w_box / w_sem: 10 * coefficient / (num_decoder_layers + 1) (loss_helper.py:312-316).
 * Outputs: objectness_label / object_assignment (b, p) i64; stats[8 heads + 7]: per head
 * (objectness, centre, heading cls, heading reg, size cls, size reg, box, semantic) then (sum
 * objectness, sum box, sum semantic, weighted total, pos_ratio, neg_ratio, the weighted total once
 * more: the word a caller may hand out as the loss tensor and edit in place);
 * grads (heads, b, c, p): d(weighted total) / d heads[h].  npos_part: b floats,
 * part: btr_gf_loss_part_floats(b, p, heads) floats of scratch. */
typedef struct {
  int b, p, k2, nh, ns, nc, heads, c, s1, n;
  float w_obj, w_box, w_sem;
  float center_delta, heading_delta, size_delta;
} btr_gf_loss_t;
int btr_gf_loss_part_floats(int b, int p, int heads);
int btr_gf_loss_fwd(const btr_gf_loss_t *d, const float *const *heads, const float *base_xyz,
                    const int *seed_inds, const int *sample_inds,
                    const long long *point_obj_mask, const long long *point_instance_label,
                    const float *center_label, const long long *heading_class_label,
                    const float *heading_residual_label, const long long *size_class_label,
                    const float *size_residual_label, const long long *sem_cls_label,
                    const float *mean_size, long long *objectness_label,
                    long long *object_assignment, float *npos_part, float *part, float *stats,
                    float *grads, btr_stream_t stream);

/* The weakly supervised form (centre labels only; loss_helper.py:416-554:
 * compute_objectness_loss_based_on_query_points_weak + compute_center_and_sem_cls_loss, smooth-L1):
 * objectness_label / object_assignment (b, p) i64 are INPUTS here (nearest labelled centre,
 * positive within 0.3 m: made by the caller); the centre term is clamp(smooth-L1 - 0.05 *
 * mean_size[size class], min = 0) per component, the box term centre + 0.1 * size class, there is
 * no heading or size-residual term (their statistics are 0, their gradients 0).  d->s1 / d->n and
 * the heading / residual deltas are not read.  Outputs as btr_gf_loss_fwd. */
int btr_gf_loss_weak_fwd(const btr_gf_loss_t *d, const float *const *heads,
                         const float *base_xyz, const long long *objectness_label,
                         const long long *object_assignment, const float *center_label,
                         const long long *size_class_label, const long long *sem_cls_label,
                         const float *mean_size, float *npos_part, float *part, float *stats,
                         float *grads, btr_stream_t stream);

/* ---- GroupFree3D: objectness of the seed points (csrc/gf_loss.hip) -----------------------------
 * reference: detection/GroupFree3D/models/loss_helper.py:17-78 (compute_points_obj_cls_loss_hard_topk)
 * with SigmoidFocalClassificationLoss (losses.py:21-81), after the labels are made:
 *   value[g] = scale * sum_{i in group g} a_i pt_i^gamma bce_i w,   grad[i] = d value[g(i)] / d x_i
 * (p = sigmoid(x), a = t alpha + (1-t)(1-alpha), pt = t(1-p) + (1-t)p, bce = max(x,0) - x t +
 * log1p(exp(-|x|)), t = label[i % period] in {0, 1}); x: groups * n logits, group g = elements
 * [g n, (g+1) n) -- one group for the seed points, one per prediction head for the query points'
 * objectness of the weakly supervised branch (loss_helper.py:416-476: the same labels for every
 * head, period = n); one launch instead of ~70 element-wise ones. */
int btr_focal_sum(int groups, int n, int period, const float *x, const long long *label, float w,
                  float scale, float gamma, float alpha, float *value, float *grad,
                  btr_stream_t stream);

/* ---- GroupFree3D: decode of one PredictHead's raw output (csrc/gf_loss.hip) --------------------
 * reference: detection/GroupFree3D/models/modules.py:233-262 and the query-position bookkeeping of
 * detector.py:204-230.  Element (b, p, ch) of the head output is out[b*sb + p*sp + ch*sc]
 * (channel order as in btr_gf_loss_fwd); base_xyz (b, p, 3); mean_size (ns, 3).
 * center (b, p, 3) = base_xyz + centre residual; heading_residuals (b, p, nh) = normalized * pi/nh;
 * size_residuals (b, p, ns, 3) = normalized * mean_size; pred_size (b, p, 3) = (size_residuals +
 * mean_size)[arg-max size class]; query_pos (b, p, 6) = (center, pred_size) and query_pos_t
 * (b, 6, p) its transpose -- the next decoder layer's position-embedding input. */
int btr_gf_head_decode(int b, int p, int nh, int ns, const float *out, long long sb, long long sp,
                       long long sc, const float *base_xyz, const float *mean_size, float *center,
                       float *heading_residuals, float *size_residuals, float *pred_size,
                       float *query_pos, float *query_pos_t, btr_stream_t stream);

/* ---- measurement hook: live time of the GEMM family (csrc/sa_mlp.hip) ---------------------------
 * Between _begin and _end every entry point of the grouped-MLP GEMM family called on this host
 * thread -- btr_sa_gemm_nt / _rc / _poolfwd / _pool, btr_sa_gemm_tn / _rc / _pool, btr_sa_bwd_fused,
 * btr_sa_bwd_gram, btr_pm_gemm_nt / _sm, the per-point first layer and the batched split-K
 * reductions, whether a script or one of the whole-layer calls issues them -- is bracketed by a HIP
 * event pair on its own stream.  _end synchronises the device, returns the summed milliseconds
 * and the number of pairs (bench.py subtracts an empty pair's cost per pair), and closes the trace. */
void btr_gemm_trace_begin(void);
int btr_gemm_trace_end(double *total_ms, int *pairs);
/* Work of the launches the last closed trace bracketed, as EXECUTED: every entry point declares the
 * flops (2 per multiply-add of its matrix products) and the operand bytes (each operand and result
 * once) of its own launches; rows of compact layers are the device's count (copied to the host on
 * the launch's stream in front of its event pair).  dense_flops: the same launches at the host's
 * row bounds (every padded neighbour a row, as in the reference's formulation). */
int btr_gemm_trace_work(double *flops, double *bytes, double *dense_flops);

/* ---- GroupFree3D: the decoder stack as one call per direction (csrc/gf_stack.hip) --------------
 * reference: detection/GroupFree3D/models/detector.py:161-219 -- the loop over the decoder layers
 *     query_pos -> self_posembed, key_pos -> cross_posembed (modules.py:50-65),
 *     decoder[i](query, key, query_pos, key_pos)            (transformer.py:36-76),
 *     prediction_heads[i](query, base_xyz)                   (modules.py:107-262),
 *     query_pos <- (center, pred_size).detach()              (detector.py:204-230)
 * -- and its autograd backward.  The host walks the same launches the per-module entry points
 * above issue (btr_pm_chain_*, btr_decoder_layer_*, btr_gf_head_decode); what disappears is the
 * interpreter between them: 30 autograd nodes per step, each ~0.1 ms of host time around ~10
 * launches of 5 - 20 us, on a step that is paced by the host.
 * Layer i: qpos[i] / kpos[i] are the position-embedding chains ((b, 3|6, pq) -> (b, e, pq) and
 * (b, 3, pk) -> (b, e, pk); has_qpos / has_kpos = 0: none), layer[i] the decoder layer, head[i] the
 * prediction head as ONE chain whose last layer is the concatenation of the seven output
 * convolutions (head_c = 4 + 2 nh + 4 ns + classes channels, order as in btr_gf_loss_fwd).
 * All chains carry need_dx as the backward needs it (head: 1, embeddings: 0). */
#define BTR_GF_MAX_DECODER_LAYERS 12
typedef struct {
  int layers, b, pq, pk, e;
  int nh, ns, head_c;
  int has_qpos, has_kpos;
  btr_decoder_layer_t layer[BTR_GF_MAX_DECODER_LAYERS];
  btr_pm_chain_t qpos[BTR_GF_MAX_DECODER_LAYERS];
  btr_pm_chain_t kpos[BTR_GF_MAX_DECODER_LAYERS];
  btr_pm_chain_t head[BTR_GF_MAX_DECODER_LAYERS];
} btr_gf_stack_t;

typedef struct {
  btr_decoder_plan_t layer[BTR_GF_MAX_DECODER_LAYERS];
  btr_pm_plan_t qpos[BTR_GF_MAX_DECODER_LAYERS], kpos[BTR_GF_MAX_DECODER_LAYERS],
      head[BTR_GF_MAX_DECODER_LAYERS];
  /* byte offsets into `saved`: the modules' own blocks, then the rows handed between them --
   * x[i] (b*pq, e) = output of layer i, qpos_cl[i] (b*pq, e) / kpos_cl[i] (b*pk, e) = embeddings */
  size_t s_layer[BTR_GF_MAX_DECODER_LAYERS], s_qpos[BTR_GF_MAX_DECODER_LAYERS],
      s_kpos[BTR_GF_MAX_DECODER_LAYERS], s_head[BTR_GF_MAX_DECODER_LAYERS];
  size_t s_x[BTR_GF_MAX_DECODER_LAYERS], s_qpos_cl[BTR_GF_MAX_DECODER_LAYERS],
      s_kpos_cl[BTR_GF_MAX_DECODER_LAYERS];
  size_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes;
  /* float offsets into `grads`: each module's block, laid out as its own plan says */
  size_t g_layer[BTR_GF_MAX_DECODER_LAYERS], g_qpos[BTR_GF_MAX_DECODER_LAYERS],
      g_kpos[BTR_GF_MAX_DECODER_LAYERS], g_head[BTR_GF_MAX_DECODER_LAYERS];
  size_t grads_floats;
} btr_gf_stack_plan_t;

/* sizeof(btr_gf_stack_t) (which = 0) / sizeof(btr_gf_stack_plan_t) (1), for bindings to check */
long long btr_gf_stack_sizeof(int which);
int btr_gf_stack_plan(const btr_gf_stack_t *d, btr_gf_stack_plan_t *plan);
/* query_cl (b*pq, e), key_cl (b*pk, e): channel-last rows; qpos0_t (b, 3|6, pq): the first
 * layer's query position, key_xyz_t (b, 3, pk): the key position (NULL without the embedding);
 * base_xyz (b, pq, 3); mean_size (ns, 3).  Per layer i (arrays of `layers` pointers):
 * head_out[i] (b, head_c, pq), head_out_cl[i] (b*pq, ceil4(head_c)), and what
 * btr_gf_head_decode derives from it: center[i], heading_residuals[i], size_residuals[i],
 * pred_size[i], query_pos[i] (b, pq, 6), query_pos_t[i] (b, 6, pq) (= layer i+1's embedding
 * input).  last_bcp (b, e, pq) / last_cl: the last layer's output (may be NULL). */
int btr_gf_stack_forward(const btr_gf_stack_t *d, const btr_gf_stack_plan_t *plan,
                         const float *query_cl, const float *key_cl, const float *qpos0_t,
                         const float *key_xyz_t, const float *base_xyz, const float *mean_size,
                         float *const *head_out, float *const *head_out_cl, float *const *center,
                         float *const *heading_residuals, float *const *size_residuals,
                         float *const *pred_size, float *const *query_pos,
                         float *const *query_pos_t, float *last_bcp, float *last_cl, void *saved,
                         void *scratch, btr_stream_t stream);
/* dhead[i] (b, head_c, pq) or NULL (that head received no gradient); dlast_bcp (b, e, pq) or NULL;
 * grads: plan->grads_floats floats, fully written; dquery_bcp (b, e, pq), dkey_bcp (b, e, pk) */
int btr_gf_stack_backward(const btr_gf_stack_t *d, const btr_gf_stack_plan_t *plan,
                          const float *query_cl, const float *key_cl,
                          const float *const *dhead, const float *dlast_bcp, void *saved,
                          float *grads, float *dquery_bcp, float *dkey_bcp, void *scratch,
                          btr_stream_t stream);
/* Replayed HIP graphs (csrc/graph_cache.hip).  btr_gf_stack_forward / _backward issue 280 / 420
 * launches of 5 - 20 us; a call whose descriptor, arguments and pointer-array contents are ALL
 * the same bytes as an earlier call's is captured into a HIP graph the second time it is seen and
 * replayed from then on (host: ~15 us per call instead of ~3 us per launch; idle time between
 * two dependent kernels ~1.1 us instead of 2.5 - 3.5).  A caller gets replays by keeping its
 * buffers and seeds (the dropout step counter lives in device memory, so a replay still draws
 * new masks); nothing else changes -- same launches, same results.  Calls inside somebody
 * else's stream capture, with the GEMM trace on, or with BTR_GRAPHS=0 are issued launch by launch.
 * A captured graph holds the raw pointers of its call: btr_graph_clear() before freeing buffers
 * that may be reused at the same address with different contents is NOT needed (the key is the
 * pointers, not the contents), but it releases the graphs' memory.
 * btr_graph_stats: calls replayed / captured / issued launch by launch since load. */
void btr_graph_stats(long long *replays, long long *captures, long long *eager);
void btr_graph_clear(void);

/* ---- Adam / AdamW over many tensors in one launch (csrc/optimizer.hip) -----------------------
 * reference: optimizer.step() of train_Votenet_FSB.py:231 (Adam) and train_GF_FSB.py:319 (AdamW,
 * two parameter groups); update rule of torch's fused implementation (header of the source).
 * items (device): per tensor its parameter, both moments, element count, `step` = the DEVICE f32
 * scalar holding this tensor's 1-based step count (already incremented for this update: what
 * torch keeps in optimizer.state[p]['step']; the bias corrections 1 - beta^step are evaluated
 * from it on the device, in double like torch's kernel, so tensors whose counts differ -- a
 * parameter without a gradient on earlier steps -- and steps taken by other kernels in between are
 * handled), `group` = index into `groups`, vec = 1 when all four pointers are 16-byte aligned and
 * n % 4 == 0;
 * grads (HOST struct, copied into the kernel arguments): the gradient pointers of tensors
 * tensor0 .. tensor0 + BTR_ADAM_MAX_TENSORS - 1; groups (HOST struct, kernel arguments): learning
 * rate and weight decay of each parameter group AT THIS STEP (a per-iteration scheduler does not
 * touch the device table); chunk_map (device, already offset to the first
 * chunk of tensor0): per workgroup (tensor index, first element), chunks of btr_adam_chunk()
 * elements; 1 - beta is formed in double on the host like torch's kernel does;
 * grad_scale: NULL or a device scalar every gradient is divided by (folded gradient clipping). */
typedef struct {
  float *p, *m, *v;
  const float *step;
  long long n;
  int group, vec;
} btr_adam_item_t;
#define BTR_ADAM_MAX_TENSORS 448
#define BTR_ADAM_MAX_GROUPS 8
typedef struct {
  const float *g[BTR_ADAM_MAX_TENSORS];
} btr_adam_grads_t;
typedef struct {
  float lr[BTR_ADAM_MAX_GROUPS], wd[BTR_ADAM_MAX_GROUPS];
} btr_adam_groups_t;
int btr_adam_chunk(void);
int btr_adam_multi(int chunks, int tensor0, const btr_adam_item_t *items,
                   const btr_adam_grads_t *grads, const btr_adam_groups_t *groups,
                   const int *chunk_map, double beta1, double beta2, double eps, int decoupled,
                   const float *grad_scale, btr_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(parameters, clip) (train_GF_FSB.py:316-318) over the same table,
 * folded into the step: btr_grad_sumsq_multi writes one sum of squares per chunk of the chunk map
 * (partial, already offset like chunk_map); btr_grad_norm_final adds all `chunks` of them in a
 * fixed order and writes out[0] = total 2-norm, out[1] = max(1, (norm + 1e-6) / clip) -- the
 * grad_scale operand of btr_adam_multi. */
int btr_grad_sumsq_multi(int chunks, int tensor0, const btr_adam_item_t *items,
                         const btr_adam_grads_t *grads, const int *chunk_map, float *partial,
                         btr_stream_t stream);
int btr_grad_norm_final(int chunks, const float *partial, float clip, float *out,
                        btr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BTR_POINTNET2_H_ */

/*
 * pointnet2_oracle.c -- CPU restatement of the reference's nine pointnet2 CUDA ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under backtoreality_amd/ may import, link or call this
 * file; it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have an
 * independent checker for the HIP path.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - The reference's native path is CUDA-only (.cu + <cuda_runtime.h>, wrappers refuse CPU
 *     tensors with "CPU not supported"), so it can be neither compiled nor run in this image:
 *     unbuildable here, no oracle/_ref.
 *   - three_interpolate(+grad): pinned by the reference's only test
 *     (detection/Votenet/pointnet2/pointnet2_test.py:18-30, reproduced in tests/).
 *   - the other ops: PARITY UNPINNED by reference tests/golden vectors (none exist); pinned
 *     solely by following the .cu sources line by line (citations on each function) and by
 *     the fixtures in tests/golden/ generated from the reference's own Python layers running
 *     over this oracle.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp -DBTR_FMAD={0,1,2} (see oracle/Makefile).
 * -ffp-contract=off is load bearing: the compiler must not fuse anything by itself; the ONE
 * expression that decides indices -- a*a + b*b + c*c of the coordinate differences -- is
 * rounded by btr_sq3() below according to BTR_FMAD, the same three modes the HIP kernels have
 * (backtoreality_amd/csrc/common.hpp), and indices are compared bit-exactly per mode:
 *   1 (default): fmaf(c,c, fmaf(a,a, b*b)) -- what nvcc -O2 (default --fmad=true, the reference's
 *      setup.py:22-25) emits for ((a*a + b*b) + c*c): mul.f32 b,b; fma.rn a,a,.; fma.rn c,c,.
 *      (NVPTX/LLVM contraction rule: an fadd fuses its FIRST fmul operand, else the second);
 *   2: fmaf(c,c, fmaf(b,b, a*a)) -- the left-to-right chain;
 *   0: as written, no contraction (an nvcc --fmad=false build).
 * Which of 1/2 a given nvcc emits cannot be checked here (no nvcc): see DESIGN.md section 2.
 *
 * All citations are relative to /root/reference/detection/Votenet/pointnet2/_ext_src/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BTR_TOTAL_THREADS 512 /* include/cuda_utils.h:18 */

#ifndef BTR_FMAD
#define BTR_FMAD 1
#endif
int btr_oracle_fmad_mode(void) { return BTR_FMAD; }

/* a*a + b*b + c*c as the reference's build rounds it (sampling_gpu.cu:105,108-109;
 * ball_query_gpu.cu:36-38; interpolate_gpu.cu:38). */
static inline float btr_sq3(float a, float b, float c) {
#if BTR_FMAD == 1
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#elif BTR_FMAD == 2
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#else
  return (a * a) + (b * b) + (c * c);
#endif
}
/* p1*w1 + p2*w2 + p3*w3 (interpolate_gpu.cu:103-104) under the same contraction rule. */
static inline float btr_dot3(float p1, float w1, float p2, float w2, float p3, float w3) {
#if BTR_FMAD == 1
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p1, w1, p2 * w2));
#elif BTR_FMAD == 2
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p2, w2, p1 * w1));
#else
  return p1 * w1 + p2 * w2 + p3 * w3;
#endif
}

/* include/cuda_utils.h:20-24 -- opt_n_threads: 2^floor(log2(work_size)) clamped to [1,512],
 * evaluated in double exactly as the reference does (the quotient of two logs). */
int btr_oracle_opt_n_threads(int work_size) {
  if (work_size <= 0) return 1;
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > BTR_TOTAL_THREADS) v = BTR_TOTAL_THREADS;
  if (v < 1) v = 1;
  return v;
}

/* include/cuda_utils.h:26-33 -- opt_block_config(x, y) -> (x_threads, y_threads). */
void btr_oracle_opt_block_config(int x, int y, int *x_threads, int *y_threads) {
  const int xt = btr_oracle_opt_n_threads(x);
  int yt = btr_oracle_opt_n_threads(y);
  if (yt > BTR_TOTAL_THREADS / xt) yt = BTR_TOTAL_THREADS / xt;
  if (yt < 1) yt = 1;
  *x_threads = xt;
  *y_threads = yt;
}

/* ------------------------------------------------------------------------------------------
 * furthest_point_sampling -- src/sampling_gpu.cu:74-178 (kernel), :180-234 (dispatch),
 * src/sampling.cpp:70-91 (temp is a (B,N) scratch pre-filled with 1e10, idxs pre-zeroed).
 *
 * Literal emulation of one thread block of `block_size` threads per batch element:
 *   - thread tid scans k = tid, tid+bs, ... keeping the first strict maximum (:100-115),
 *     starting from best=-1, besti=0 (:95-96); points with x^2+y^2+z^2 <= 1e-3 are skipped
 *     (never updated, never selected) (:105-106; the float sum is compared against the
 *     double literal 1e-3);
 *   - shared-memory tree reduction, stride bs/2 .. 1, where slot t absorbs slot t+stride and
 *     keeps its own index on ties (`v2 > v1 ? i2 : i1`, :64-70, :121-174).
 * temp is updated in place exactly as the kernel does.  block_size <= 0 selects
 * opt_n_threads(n) like the dispatcher (:182).
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_furthest_point_sampling_bs(int b, int n, int m, const float *dataset,
                                           float *temp, int *idxs, int block_size) {
  if (m <= 0 || b <= 0) return; /* :78 */
  if (block_size <= 0) block_size = btr_oracle_opt_n_threads(n);
  const int bs = block_size;
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    float *dists = (float *)malloc(sizeof(float) * bs);
    int *dists_i = (int *)malloc(sizeof(int) * bs);
    int old = 0;
    out[0] = old; /* :91-92 */
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += bs) {
          const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
          const float mag = btr_sq3(x2, y2, z2);
          if (mag <= 1e-3) continue;
          const float d = btr_sq3(x2 - x1, y2 - y1, z2 - z1);
          const float d2 = d < tp[k] ? d : tp[k]; /* min(d, temp[k]) */
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int stride = bs / 2; stride >= 1; stride >>= 1) {
        for (int tid = 0; tid < stride; ++tid) { /* __update(dists, dists_i, tid, tid+stride) */
          const float v1 = dists[tid], v2 = dists[tid + stride];
          const int i1 = dists_i[tid], i2 = dists_i[tid + stride];
          dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
}

void btr_oracle_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                        int *idxs) {
  btr_oracle_furthest_point_sampling_bs(b, n, m, dataset, temp, idxs, 0);
}

/* Closed form of the same selection rule, used by tests to cross-check the emulation above:
 * the winner is the maximum of d2 under the total order (d2 descending, then
 * bitreverse_{log2 bs}(k mod bs) ascending, then k ascending).  Skipped points never compete;
 * if nothing competes the result is index 0 (best=-1, besti=0 everywhere). */
static unsigned btr_bitrev(unsigned v, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

void btr_oracle_furthest_point_sampling_closed_form(int b, int n, int m, const float *dataset,
                                                    float *temp, int *idxs, int block_size) {
  if (m <= 0 || b <= 0) return;
  if (block_size <= 0) block_size = btr_oracle_opt_n_threads(n);
  int bits = 0;
  while ((1 << bits) < block_size) ++bits;
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      float best = -1;
      int besti = 0;
      unsigned long long bestkey = ~0ull;
      for (int k = 0; k < n; ++k) {
        const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
        const float mag = btr_sq3(x2, y2, z2);
        if (mag <= 1e-3) continue;
        const float d = btr_sq3(x2 - x1, y2 - y1, z2 - z1);
        const float d2 = d < tp[k] ? d : tp[k];
        tp[k] = d2;
        const unsigned long long key =
            ((unsigned long long)btr_bitrev((unsigned)(k % block_size), bits) << 32) |
            (unsigned)k;
        if (d2 > best || (d2 == best && key < bestkey)) {
          best = d2;
          besti = k;
          bestkey = key;
        }
      }
      old = besti;
      out[j] = old;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * gather_points / gather_points_grad -- src/sampling_gpu.cu:13-25, :39-52.
 * out[b,c,j] = points[b,c,idx[b,j]];  grad_points[b,c,idx[b,j]] += grad_out[b,c,j]
 * (grad_points pre-zeroed, src/sampling.cpp:56-58).  The reference accumulates with
 * atomicAdd (order unspecified); the oracle accumulates in ascending j.
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_gather_points(int b, int c, int n, int npoints, const float *points,
                              const int *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j) {
        const int a = idx[(size_t)i * npoints + j];
        out[((size_t)i * c + l) * npoints + j] = points[((size_t)i * c + l) * n + a];
      }
}

void btr_oracle_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                   const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j) {
        const int a = idx[(size_t)i * npoints + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * npoints + j];
      }
}

/* ------------------------------------------------------------------------------------------
 * ball_query -- src/ball_query_gpu.cu:14-49; idx pre-zeroed (src/ball_query.cpp:24-26).
 * For each centre scan k ascending, take the first nsample points with d2 < radius^2
 * (strict, f32, radius2 = radius*radius in f32 :27); the first hit fills all nsample slots
 * (:39-43); no hit leaves the row all-zero.
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                           const float *xyz, int *idx) {
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *px = xyz + (size_t)bi * n * 3;
      const float *pc = new_xyz + ((size_t)bi * m + j) * 3;
      int *row = idx + ((size_t)bi * m + j) * nsample;
      const float new_x = pc[0], new_y = pc[1], new_z = pc[2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = px[k * 3 + 0], y = px[k * 3 + 1], z = px[k * 3 + 2];
        const float d2 = btr_sq3(new_x - x, new_y - y, new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * group_points / group_points_grad -- src/group_points_gpu.cu:13-33, :48-69.
 * out[b,l,j,k] = points[b,l,idx[b,j,k]];  grad_points[b,l,idx[b,j,k]] += grad_out[b,l,j,k]
 * (pre-zeroed, src/group_points.cpp:52-54; atomicAdd in the reference, ascending (j,k) here).
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                             const int *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * npoints * nsample;
      float *o = out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) o[j * nsample + k] = p[id[j * nsample + k]];
    }
}

void btr_oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                  const float *grad_out, const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *gp = grad_points + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * npoints * nsample;
      const float *go = grad_out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) gp[id[j * nsample + k]] += go[j * nsample + k];
    }
}

/* ------------------------------------------------------------------------------------------
 * three_nn -- src/interpolate_gpu.cu:14-64.  best1..3 are doubles initialised to 1e40 (:32);
 * the f32 distance is compared after promotion with strict `<` (:39-54) so the earliest index
 * wins ties; fewer than three known points leave 1e40 -> +inf after the float store and
 * index 0.  Outputs are SQUARED distances (the Python wrapper takes the sqrt,
 * pointnet2_utils.py:140-142).
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_three_nn(int b, int n, int m, const float *unknown, const float *known,
                         float *dist2, int *idx) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *u = unknown + ((size_t)bi * n + j) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        const float d = btr_sq3(ux - x, uy - y, uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d;     besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d;     besti2 = k;
        } else if (d < best3) {
          best3 = d;     besti3 = k;
        }
      }
      float *d2 = dist2 + ((size_t)bi * n + j) * 3;
      int *id = idx + ((size_t)bi * n + j) * 3;
      d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3;
      id[0] = besti1; id[1] = besti2; id[2] = besti3;
    }
}

/* ------------------------------------------------------------------------------------------
 * three_interpolate / three_interpolate_grad -- src/interpolate_gpu.cu:77-106, :121-148.
 * out[b,l,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3 evaluated left to right in f32 (:103-104);
 * grad: three atomicAdds of grad_out*w per output element (:144-146), ascending j here.
 * ------------------------------------------------------------------------------------------ */
void btr_oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                                  const int *idx, const float *weight, float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * m;
      const int *id = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        o[j] = btr_dot3(p[i1], w1, p[i2], w2, p[i3], w3);
      }
    }
}

void btr_oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                       const int *idx, const float *weight,
                                       float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *gp = grad_points + ((size_t)bi * c + l) * m;
      const int *id = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      const float *go = grad_out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        gp[i1] += go[j] * w1;
        gp[i2] += go[j] * w2;
        gp[i3] += go[j] * w3;
      }
    }
}

/* tensorf_hip.h — C ABI of libtensorf_hip.so (MI355X / gfx950).
 *
 * The reference (hautran7201/3D-reconstruction) has no FFI layer: its hot path is Python calling
 * ATen eager ops.  This header is the drop-in boundary a HIP replacement exports:
 * plain pointers + sizes, no torch types, no allocation inside, every entry point enqueues on the
 * caller's stream and returns a hipError_t as int.  Each entry cites the reference code it replaces
 * (paths relative to the reference repo).
 *
 * Memory layout (HBM), all fp32 unless noted:
 *   rays            (R,6) row-major [ox oy oz dx dy dz]                       renderer.py:18
 *   plane i         [H=G[mat1]][W=G[mat0]][C_i]  channel-LAST view of the reference's
 *                   (1,C_i,H,W) parameter (torch channels_last strides, zero copy)  tensoRF.py:157-158
 *   line i          [G[vec]][C_i]                channel-last view of (1,C_i,G,1)   tensoRF.py:159-160
 *   alpha cells     uint8 [(Gz+1)][(Gy+1)][(Gx+1)], one byte per trilinear cell: bit (dz*4+dy*2+dx) =
 *                   alpha_volume[z0+dz][y0+dy][x0+dx] > 0, cell index = floor coord + 1 (so -1..G-1)
 *                   — the exact `grid_sample(...) > 0` predicate of tensorBase.py:41-45,350-351 in 1 byte
 *   packed app list one entry per shaded sample (weight > rayMarch_weight_thres), a ray's entries
 *                   contiguous and in sample order: app_ray[s], app_xyz[s][3] (normalised), app_w[s]
 */
#ifndef TENSORF_HIP_H
#define TENSORF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tf_stream_t; /* hipStream_t */

/* The packed app list is written through TF_N_SHARDS independent reservation counters (one per
 * 128-B line, TF_SHARD_STRIDE ints apart) so that one atomic per ray never serialises on a single
 * word.  Shard g owns packed entries [g*seg_cap, g*seg_cap + counters[g*TF_SHARD_STRIDE]) with
 * seg_cap = TfMarchIO.seg_cap entries per shard (0: the worst case ceil(R / TF_N_SHARDS) * N).  counters[g*STRIDE+1] / [+2]
 * accumulate the number of density samples / in-bbox samples (statistics for the roofline report).
 * Right-sized lists: a ray whose samples do not fit its shard's remaining room writes the part that fits (every position
 * stays inside the buffers), the counters keep counting the DEMAND (so the host knows how much room the batch needs),
 * every reader takes min(counter, seg_cap), and counters[TF_OVERFLOW_SLOT] is set: the step's results are then
 * incomplete — tf_composite_forward publishes the flag (TfLive), tf_adam_step refuses to update, the host re-runs the
 * batch with a larger workspace. */
#define TF_N_SHARDS 64
#define TF_SHARD_STRIDE 32
#define TF_MAX_SAMPLES 8192   /* samples per ray handled by one LDS queue */
#define TF_TILE 64            /* shaded samples per shading tile */
#define TF_TICKET_SLOT 4      /* counters[4]: tile ticket of tf_shade_forward (zeroed with the counters) */
#define TF_OVERFLOW_SLOT 5    /* counters[5]: non-zero once a packed list (bit 0: shaded samples, bit 1: density entries) ran full */

enum { TF_MODEL_VM = 0, TF_MODEL_CP = 1 };
enum { TF_ACT_SOFTPLUS = 0, TF_ACT_RELU = 1 };
enum { TF_HEAD_MLP = 0, TF_HEAD_SH = 1, TF_HEAD_RGB = 2 };
enum { TF_SRC_FEAT = 0, TF_SRC_VIEW = 1, TF_SRC_PTS = 2 };

/* One decomposition (density or appearance).  TensorVMSplit: 3 planes + 3 lines, component counts
 * n_comp[i].  TensorCP: plane[] = NULL, 3 lines of n_comp[0] components.  mask[i] is the per-component
 * FreeNeRF decomposition mask (C_i floats) or NULL.   tensoRF.py:207-263, 358-415 */
typedef struct TfFactors {
    const float* plane[3];
    const float* line[3];
    const float* mask[3];
    int n_comp[3];
} TfFactors;

/* Gradient destinations, same layouts as TfFactors (accumulated with float atomics).  Line tensors are
 * tiny (G x C) and every ray hits them, so their atomics are spread over `n_rep` replicas
 * (replica r of line i at line[i] + r*rep_stride floats, workgroup b uses replica b % n_rep); the caller
 * sums the replicas afterwards with tf_reduce_replicas. */
typedef struct TfFactorGrads {
    float* plane[3];
    float* line[3];
    int n_rep;        /* >= 1 */
    int rep_stride;   /* floats between consecutive replicas */
} TfFactorGrads;

/* Geometry + density field + alpha mask: everything TensorBase.forward reads before shading.
 * tensorBase.py:52-116 (ctor / update_stepSize), :30-48 (AlphaGridMask). */
typedef struct TfField {
    int model;               /* TF_MODEL_* */
    int act;                 /* TF_ACT_*  (fea2denseAct, tensorBase.py:291-295) */
    int grid[3];             /* Gx, Gy, Gz */
    float aabb_lo[3], aabb_hi[3];
    float inv_aabb[3];       /* 2/(hi-lo), fp32 as tensorBase.py:106 */
    float near_, far_, step; /* near_far, stepSize (tensorBase.py:109) */
    float distance_scale, density_shift, weight_thres;
    TfFactors density;
    const uint8_t* alpha_cells; /* NULL = no alphaMask */
    int alpha_grid[3];          /* Gx, Gy, Gz of the alpha volume */
    float alpha_lo[3];
    float alpha_inv[3];         /* 1/(hi-lo)*2, fp32 as tensorBase.py:37 */
} TfField;

/* Inputs/outputs of the ray-march kernel (sampling + masks + density + transmittance scan). */
typedef struct TfMarchIO {
    const float* rays;     /* (R,6) */
    int n_rays, n_samples;
    int ndc;               /* 0: sample_ray (tensorBase.py:189-208)  1: sample_ray_ndc (:178-187) */
    const float* jitter;   /* AABB mode: (R) per-ray stratified jitter in [0,1) or NULL (eval) */
    const float* z_table;  /* NDC mode: (N) z values shared by all rays (linspace [+jitter]) */
    int save_valid;        /* 1: keep per-ray valid-sample lists for the backward pass */
    float t_stop;          /* stop marching a ray once transmittance < t_stop (0 = never) */
    /* per-ray outputs */
    float* acc;            /* (R) sum of weights                      tensorBase.py:377 */
    float* depth;          /* (R) sum w*z + (1-acc)*rays[:,-1]        tensorBase.py:386-388 */
    int* app_offset;       /* (R) first packed app entry of the ray */
    int* app_count;        /* (R) number of shaded samples of the ray */
    int* val_count;        /* (R) number of alpha-mask-surviving samples processed */
    /* sharded counters (TF_N_SHARDS*TF_SHARD_STRIDE ints), zeroed by the caller before launch */
    int* counters;
    /* packed app list (capacity TF_N_SHARDS * ceil(R/TF_N_SHARDS) * N entries) */
    int* app_ray;
    float* app_xyz;        /* (S,3) normalised coordinates */
    float* app_w;          /* (S) */
    /* saved valid lists, dense-strided [r*N + k], only when save_valid */
    int* val_idx;          /* sample index along the ray */
    float* val_feat;       /* density feature before activation */
    /* optional debug bitmaps (R * ceil(N/64) uint64 words) or NULL */
    uint64_t* dbg_bbox_bits;
    uint64_t* dbg_valid_bits;
    uint64_t* dbg_app_bits;   /* must be zeroed by the caller */
    /* training with the binned scatter (optional, with save_valid): the forward already reserves the ray's block of
     * the density entry list (counter slot 3) and writes the entries' normalised coordinates, so that the entries can
     * be sorted (tf_binned_scatter stage 1) while the shading kernels run; tf_march_backward then only fills in
     * dL/df at ent_offset[r] + k. */
    float* ent_xyz;        /* (cap,3) or NULL */
    int* ent_offset;       /* (R) first density entry of the ray */
    float* dbg_z;          /* tests (optional): (R, n_samples) the sample positions z of tensorBase.py:198-203 / :181-183 */
    int seg_cap;           /* entries per shard of the packed app list (app_ray / app_xyz / app_w and everything indexed like
                            * them: colours, saved rows); 0 = worst case ceil(R / TF_N_SHARDS) * N */
    int ent_seg_cap;       /* the same for the density entry list (ent_xyz / ent_df, counter slot 3); 0 = worst case */
} TfMarchIO;

/* One positional-encoding block of the MLP input (mlp.py:8-13, 41-66, 84-103, 126-153). */
typedef struct TfPeBlock {
    int src;               /* TF_SRC_* */
    int freqs;
    const float* mask;     /* 2*dim*freqs floats or NULL */
} TfPeBlock;

/* Appearance field + shading head. */
typedef struct TfShade {
    int model;             /* TF_MODEL_* */
    int grid[3];
    TfFactors app;
    int app_dim;
    int n_app_total;       /* sum of n_comp (VM) or n_comp[0] (CP) */
    int head;              /* TF_HEAD_* */
    const float* basis;    /* packed basis_mat: [roundup16(app_dim)][kpad(n_app_total)]   tensoRF.py:149,263; training:
                            * followed by >= 64 readable floats (tf_shade_backward reads whole 64-column groups) */
    int n_pe;
    TfPeBlock pe[3];
    int in_c;              /* MLP input width (mlp.py:31,75,113) */
    int feature_c;         /* hidden width: 64, 128 or 256 */
    const float* w1;       /* packed [feature_c][kpad(in_c)] */
    const float* b1;
    const float* w2;       /* packed [feature_c][kpad(feature_c)] */
    const float* b2;
    const float* w3;       /* [3][feature_c] */
    const float* b3;
    const float* w1t;      /* training only: packed transpose [kpad(in_c)][feature_c] of w1 */
    const float* w2t;      /* training only: packed transpose [feature_c][feature_c] of w2 */
} TfShade;

/* kpad(k) = round k up to a multiple of 16 (row stride of every packed matrix = 4 MFMA k-steps). */

/* Builds the 1-byte-per-cell occupancy table from an alpha volume (Gz,Gy,Gx) of non-negative floats.
 * Replaces the 8-tap trilinear grid_sample of AlphaGridMask.sample_alpha, tensorBase.py:41-45. */
int tf_pack_alpha_cells(const float* volume, int gx, int gy, int gz, uint8_t* cells, tf_stream_t stream);

/* Zero-pads a row-major (rows, cols) matrix into (rows_pad, kpad(cols)).  Used for basis_mat and the
 * MLP weights (tensoRF.py:149, mlp.py:34-36). */
int tf_pack_matrix(const float* src, int rows, int cols, float* dst, int rows_pad, tf_stream_t stream);
/* dst[c][r] = src[r][c], dst is (kpad(cols), rows_pad) zero padded: the transposes the backward GEMMs read. */
int tf_pack_matrix_t(const float* src, int rows, int cols, float* dst, int rows_pad, tf_stream_t stream);
/* Several of those copies in ONE launch (the training step refreshes basis, w1, w2, w1^T, w2^T after every
 * optimizer step): item k writes dst = padded copy of src (rows x cols), or its transpose. */
#define TF_PACK_MAX 8
typedef struct TfPackItem {
    const float* src;
    float* dst;
    int rows, cols, rows_pad, transpose;
} TfPackItem;
typedef struct TfPackJob {
    int n;
    int n_zero;               /* > 0: the launch also zeroes `zero[0..n_zero)` (the forward's counter / histogram block) */
    TfPackItem item[TF_PACK_MAX];
    int* zero;
} TfPackJob;
int tf_pack_matrices(const TfPackJob* job, tf_stream_t stream);

/* The photometric loss of train.py:334 and its gradient in one launch: *loss = mean((a - b)^2) over n floats,
 * grad[i] = grad_scale * 2 (a[i] - b[i]) / n.  grad_scale is 1 for a single process and 1 / world_size under data
 * parallelism, where the ranks' gradients are then SUMMED (no separate averaging pass over the gradient buffer).
 * (Used by the hipGraph-captured step; eager callers keep their torch expression.) */
int tf_mse_grad(const float* a, const float* b, int n, float grad_scale, float* loss, float* grad, tf_stream_t stream);

/* sample_ray / sample_ray_ndc + bbox test + AlphaGridMask test + compute_densityfeature +
 * feature2density + raw2alpha + app_mask + acc/depth reductions:
 * tensorBase.py:178-208, 339-370, 377, 386-388; tensoRF.py:207-227, 358-386. */
int tf_march_forward(const TfField* field, const TfMarchIO* io, tf_stream_t stream);

/* compute_appfeature + renderModule on the packed app list: tensoRF.py:230-263, 388-415; mlp.py.
 * counters/seg_cap describe the sharded packed list; rays gives view directions (normalised when ndc,
 * tensorBase.py:343); rgb_out is (capacity,3), written at the packed positions.  The kernel is persistent (tiles
 * are handed out by a ticket): max_workgroups (0 = all 512 slots, two per CU) leaves CU slots free for kernels the
 * caller runs next to it on another stream (the training step's early sorts of the binned scatter). */
typedef struct TfShadeSave {  /* training: rows kept for tf_shade_backward (both NULL in inference) */
    float* x;              /* (cap, kpad(in_c)): MLP input [feat, view, PE blocks] of every packed sample, zero padded */
    float* v;              /* (cap, n_app_total): plane*line products (the operand of basis_mat, tensoRF.py:263) */
    float* h1;             /* (cap, feature_c): first hidden layer after ReLU  (mlp.py:34-35) */
    float* h2;             /* (cap, feature_c): second hidden layer after ReLU (mlp.py:36-37) */
} TfShadeSave;
int tf_shade_forward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                     const int* app_ray, const float* app_xyz, float* rgb_out, int max_workgroups,
                     const TfShadeSave* save, tf_stream_t stream);
/* Which kernel tf_shade_forward launches: 0 (default) = the pipelined one-workgroup-per-CU kernel where its conditions
 * hold (MLP head, feature_c = 128, its LDS layout fits) and the two-workgroups-per-CU kernel elsewhere; 1 = the latter
 * everywhere.  Process-wide; the parity tests run both. */
int tf_shade_forward_variant(int variant);

/* rgb_map = sum w*rgb (+ 1-acc when bg) clamped to [0,1]: tensorBase.py:378-384.  rgb_pre (optional)
 * receives the pre-clamp value, which the backward needs for the clamp mask.  n_shaded (optional, with the sharded
 * counters of the march kernel) receives num_valid_samples = app_mask.sum() (tensorBase.py:390) as one int64. */
/* `live` (optional): the step's sample counts for the optimizer's gates (TfAdamJob.live) and for the host: dev[0] =
 * number of density samples (ray_valid.sum(), tensorBase.py:359), dev[1] = number of shaded samples (app_mask.sum(),
 * :370) as floats in device memory; host[0..1] the same as ints in PINNED host memory (read them after an event
 * recorded behind this launch: the autograd binding returns None gradients for the tensors the reference's graph would
 * not contain).  dev[2] / host[2]: non-zero when a right-sized list overflowed (TF_OVERFLOW_SLOT): the step is incomplete.
 * Either pointer may be NULL. */
typedef struct TfLive {
    float* dev;            /* 3 floats: density samples, shaded samples, overflow flags (counters[TF_OVERFLOW_SLOT]) */
    int* host;             /* n_slots x 4 ints in pinned host memory: the same three (+ 1 spare) per slot */
    const float* slot;     /* NULL (slot 0), or a device float holding the slot this step reports to — a captured launch
                            * cannot change its arguments, the caller changes *slot (graph.GraphedTrainStep stages it) */
    int n_slots;
    int pad_;
} TfLive;
int tf_composite_forward(int n_rays, const int* app_offset, const int* app_count, const float* app_w,
                         const float* rgb, const float* acc, int white_bg, float* rgb_map, float* rgb_pre,
                         const int* counters, long long* n_shaded, const TfLive* live, tf_stream_t stream);
/* The same launch followed, inside the kernel, by the photometric loss of train.py:334 and its gradient (what
 * tf_mse_grad computes in a launch of its own): grad[i] = grad_scale * 2 (rgb_map[i] - target[i]) / (3 n_rays), and
 * *loss = mean((rgb_map - target)^2), written once by the last workgroup to finish.  `state` = 2 device words, zero when
 * first used; the kernel leaves them zero.  (The captured training step uses this form: one launch less on its critical
 * path.) */
typedef struct TfLossFuse {
    const float* target;   /* (n_rays, 3) */
    float grad_scale;
    float* grad;           /* (n_rays, 3) out */
    float* loss;           /* 1 float out */
    float* state;          /* 2 words: running sum, arrival counter */
} TfLossFuse;
int tf_composite_forward_loss(int n_rays, const int* app_offset, const int* app_count, const float* app_w,
                              const float* rgb, const float* acc, int white_bg, float* rgb_map, float* rgb_pre,
                              const int* counters, long long* n_shaded, const TfLossFuse* fuse, const TfLive* live,
                              tf_stream_t stream);

/* compute_densityfeature / compute_appfeature on an explicit point list (normalised coordinates),
 * the public hooks used by compute_alpha (tensorBase.py:298-318): out_f (S) / out_feat (S, app_dim). */
int tf_density_points(const TfField* field, const float* xyz_n, int n, float* out_f, tf_stream_t stream);
int tf_appfeature_points(const TfShade* shade, const float* xyz_n, int n, float* out_feat, tf_stream_t stream);
/* renderModule(pts, viewdirs, features, mask) on explicit lists — the shading heads as stand-alone calls
 * (MLPRender_Fea / MLPRender_PE / MLPRender mlp.py:41-69, 84-107, 126-155; SHRender / RGBRender :15-25): pts_n (n,3)
 * normalised points (only the heads with a position encoding read them), viewdirs (n,3), features (n, app_dim);
 * rgb_out (n,3).  The encoding masks are the TfPeBlock masks of `shade`. */
int tf_shade_points(const TfShade* shade, const float* pts_n, const float* viewdirs, const float* features, int n,
                    float* rgb_out, tf_stream_t stream);

/* Backward of compositing + density: consumes d(loss)/d(rgb_map), the saved valid lists and
 * the per-sample rgb; produces d(loss)/d(rgb sample) for the shading backward and scatter-adds the
 * density factor gradients. */
int tf_march_backward(const TfField* field, const TfMarchIO* io, const float* grad_rgb_map, const float* rgb_pre,
                      int white_bg, const float* rgb, float* grad_rgb, const TfFactorGrads* dgrads,
                      float* ent_xyz, float* ent_df, tf_stream_t stream);
/* ent_xyz/ent_df != NULL: instead of scattering, the kernel appends one entry (normalised xyz, dL/df) per
 * density sample with a non-zero gradient to the sharded entry list (counter slot 3) for tf_binned_scatter.
 * When io->ent_offset is set the forward has already placed the entries (TfMarchIO.ent_xyz): only ent_df is written. */

/* Backward of the shading head + appearance lookup (autograd of tensoRF.py:230-263, mlp.py:27-155) on the rows the
 * training forward saved (TfShadeSave: MLP inputs, both hidden layers and the plane*line products; the colours are
 * the forward's rgb_out) — nothing of the forward is recomputed.  The gradients of b1, b2, w3, b3 and of the appearance
 * factors are ADDED to what the buffers hold (zero them first); those of w1, w2 and basis are WRITTEN (the fold of the
 * workgroups' slabs, wslab).  Gradient matrices use the reference's own (unpadded, row-major) layouts. */
typedef struct TfShadeGrads {
    float* w1; float* b1; float* w2; float* b2; float* w3; float* b3;
    float* basis;          /* (app_dim, n_app_total) */
    TfFactorGrads app;
    float* dv_out;         /* (cap, n_app_total): on entry the V rows saved by the forward (TfShadeSave.v), on exit
                            * the dL/dV rows of the packed samples, consumed by the scatter stage */
    float* wslab;          /* tf_shade_backward_wslab_floats() floats: per-workgroup weight-gradient slabs */
    int direct_scatter;    /* 1: scatter dv_out with per-tap atomics inside the call (TensorCP / binning disabled);
                            * 0: the caller runs tf_binned_scatter on dv_out */
    const float* x_saved;  /* (cap, kpad(in_c)): MLP input rows saved by the forward (TfShadeSave.x) */
    const float* h1_saved; /* (cap, feature_c): TfShadeSave.h1 */
    const float* h2_saved; /* (cap, feature_c): TfShadeSave.h2 */
    const float* rgb_fwd;  /* (cap, 3): the forward's rgb_out (sigmoid outputs) */
} TfShadeGrads;
size_t tf_shade_backward_wslab_floats(const TfShade* shade);
/* 1 when tf_shade_backward supports this head (MLP, feature_c 64 / 128, app_dim <= 32, in_c <= 192, and the
 * 64-sample tile fits the 160 KB of LDS; when V, X, H1, H2 do not fit side by side — more than 176 appearance
 * components at feature_c 128 — V shares the hidden layers' space and is gathered twice: up to 384 components). */
int tf_shade_backward_supported(const TfShade* shade);
int tf_shade_backward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                      const int* app_ray, const float* app_xyz, const float* grad_rgb, const TfShadeGrads* grads,
                      tf_stream_t stream);

/* ---- callers of the density lookup outside the per-ray march ---- */
/* compute_alpha on world-space points: alpha mask test, normalise, density, activation, 1-exp(-sigma*length).
 * tensorBase.py:298-318 (used by getDenseAlpha / updateAlphaMask :215-256). */
int tf_alpha_points(const TfField* field, const float* xyz, int n, float length, float* out_alpha, tf_stream_t stream);
/* getDenseAlpha (tensorBase.py:215-230) on the device: alpha of every node of the (gx, gy, gz) lattice of the field's
 * box — node (ix, iy, iz) at aabb_lo (1 - s) + aabb_hi s, s = lin_x[ix] / lin_y[iy] / lin_z[iz] (the caller's
 * torch.linspace(0, 1, G) tables, device pointers) — written as out_alpha[iz][iy][ix], the layout updateAlphaMask
 * continues with (:236-237).  No (G^3, 3) point list is built or copied. */
int tf_alpha_lattice(const TfField* field, const float* lin_x, const float* lin_y, const float* lin_z, int gx, int gy, int gz,
                     float length, float* out_alpha, tf_stream_t stream);
/* updateAlphaMask's tail (tensorBase.py:236-254): clamp, 3^3 max-pool, threshold -> volume (gz, gy, gx) of 0 / 1, and
 * stats[7] = {kept voxels, min ix, iy, iz, max ix, iy, iz} of the kept voxels (preset to {0, INT_MAX x3, -1 x3}). */
int tf_alpha_pool_threshold(const float* alpha, int gx, int gy, int gz, float thres, float* volume, int* stats,
                            tf_stream_t stream);
/* AlphaGridMask.sample_alpha: trilinear grid_sample of the float volume (Gz,Gy,Gx), align_corners, zero padding;
 * lo / inv = the mask's aabb[0] and 2/aabbSize.  tensorBase.py:41-48. */
int tf_sample_alpha_points(const float* volume, int gx, int gy, int gz, const float lo[3], const float inv[3],
                           const float* xyz, int n, float* out, tf_stream_t stream);
/* filtering_rays: keep[r] = slab test t_max > t_min (bbox_only) or "any of the N eval samples hits the alpha
 * mask".  tensorBase.py:259-288. */
int tf_filter_rays(const TfField* field, const float* rays, int n_rays, int bbox_only, int n_samples, uint8_t* keep,
                   tf_stream_t stream);

/* ---- on-device ray generation ----------------------------------------------------------
 * rays_out[t] = (origin, direction) of pixel pixel_ids[t] (or first_pixel + t when pixel_ids is NULL), pixels
 * numbered row-major j * width + i.  Camera-space direction ((i + 0.5 - cx) / fx, (j + 0.5 - cy) / fy, 1)
 * (dataLoader/ray_utils.py:24-42) or, with opengl, (.., -(..), -1) (:45-63); optionally normalised
 * (dataLoader/blender.py:59); rotated by the 3x4 camera-to-world matrix (row-major c2w[12], ray_utils.py:66-87);
 * optionally projected to NDC with near plane ndc_near (ray_utils.py:90-107). */
typedef struct TfCamera {
    int height, width;
    float fx, fy, cx, cy;
    float c2w[12];
    int opengl;
    int normalize;
    int ndc;
    float ndc_near;
} TfCamera;
/* The batch of a training step, `allrays[ray_idx]` and `allrgbs[ray_idx]` (train.py:297-298), gathered on the device
 * in one launch: rays (n_all,6), rgbs (n_all,3), ids (n) int64 row numbers -> rays_out (n,6), rgbs_out (n,3).
 * Negative ids count from the end; rows whose id is out of range are left untouched. */
int tf_gather_batch(const float* rays, const float* rgbs, long long n_all, const long long* ids, int n, float* rays_out,
                    float* rgbs_out, tf_stream_t stream);
/* The same launch also moving n_extra floats extra_src -> extra_dst: the step's host-drawn numbers (the per-ray sampling
 * jitter of tensorBase.py:198-203, drawn from the CPU generator like the reference does).  extra_src may be pinned host
 * memory (read in place by the kernel: no copy launch of its own); the caller keeps it alive until the launch has run.
 * `pack` (optional): a tf_pack_matrices job run by extra workgroups of the same launch — the captured training step's
 * weight copies and zeroed counter block, which then need no launch inside the graph. */
int tf_gather_batch_staged(const float* rays, const float* rgbs, long long n_all, const long long* ids, int n, float* rays_out,
                           float* rgbs_out, const float* extra_src, float* extra_dst, int n_extra, const TfPackJob* pack,
                           tf_stream_t stream);
int tf_generate_rays(const TfCamera* cam, const long long* pixel_ids, long long first_pixel, int n, float* rays_out,
                     tf_stream_t stream);

/* Binned ("owner computes") scatter of the VM factor gradients — the backward of the plane x line lookups
 * (autograd of tensoRF.py:216-225 / :240-260) without one global atomic per tap.
 * Global float atomics on random 64-B pieces are request-bound (~0.4 TB/s measured), so the samples are first
 * counting-sorted by destination: per plane i by the T x T texel tile holding their footprint base, per line i by
 * the bucket of LB entries holding their base entry.  One workgroup then owns a (tile | bucket) chunk,
 * accumulates all its samples in LDS ((T+1)^2 x C or (LB+1) x C floats) and flushes the block once with
 * contiguous atomics.  Entries live in the sharded layout of the packed app list:
 * shard g = [g*seg_cap, g*seg_cap + counters[g*TF_SHARD_STRIDE + slot]). */
typedef struct TfBinJob {
    int model;                /* TF_MODEL_VM, or TF_MODEL_CP (line keys only; a line's gradient is dL/d(product) times
                               * the other two lines' values; `grad` rows hold factors.n_comp[0] columns) */
    TfFactors factors;        /* the field being differentiated (density or appearance) */
    TfFactorGrads grads;
    int grid[3];
    const int* counters;      /* sharded entry counters */
    int slot;                 /* which counter of a shard holds the entry count */
    int seg_cap;
    const float* xyz;         /* (cap,3) normalised coordinates of the entries */
    const float* grad;        /* density: (cap) dL/df per entry; appearance: (cap, ld) dL/dV rows */
    int grad_ld;              /* 0: one scalar per entry, broadcast over components; else row stride */
    int tile, bucket, chunk;  /* T, LB, max entries per workgroup */
    /* workspace (ints): hist[nkeys] (read, then left as is), offsets[nkeys+1], cursor[nkeys], chunk_off[nkeys+1] followed by the
     * work-item table (16-byte aligned; one int4 per item: {key | group << 20, first position in binned[], end position, 0},
     * at most groups*nkeys + kpe*entries/chunk items, groups = most 16-component groups of a plane / line);
     * binned[kpe*entries], kpe = tf_bin_keys_per_entry */
    int* hist; int* offsets; int* cursor; int* chunk_off; int* binned;
    int nkeys;
    int hist_zeroed;          /* 1: the caller has zeroed hist[0..nkeys) on this stream (saves a launch) */
    int stage;                /* 0: sort the entries by key, then scatter; 1: sort only (needs xyz and the counters, not
                               * grad: can run as soon as the entry coordinates exist, on another stream); 2: scatter
                               * only, the workspace holds the result of an earlier stage-1 call of the same job */
    int binned_cap;           /* ints in binned[]; */
    int items_cap;            /* int4 ITEMS the work-item table behind chunk_off[nkeys + 1] holds */
    int share_groups;         /* 1: one sort key per (sample, plane | line) — the 16-component groups share it and the work-item
                               * table carries the group (cheapest sort: eager steps, large configurations); 0: one key per
                               * group (what runs best beside tf_shade_forward on the second stream: the captured step) */
    int* status;              /* device word (sticky, may be NULL): the kernels OR a TF_BIN_ERR_* bit into it instead of
                               * writing / reading outside binned[], the item table or the entry list — a histogram that
                               * does not describe the entries (e.g. not zeroed) then costs a wrong gradient and this
                               * flag, never a wild access.  Read it with tf_bin_status. */
} TfBinJob;
#define TF_BIN_ERR_BINNED 1   /* a sorted position fell outside binned[] */
#define TF_BIN_ERR_ITEMS 2    /* the work-item table would overflow */
#define TF_BIN_ERR_ENTRY 4    /* binned[] held an entry index outside the entry list */
/* Copies *status to the host (synchronises the stream): 0, or hipErrorAssert (710) with the bits in *bits_out. */
int tf_bin_status(const int* status, int* bits_out, tf_stream_t stream);
#define TF_BIN_MAX_KEYS 262144  /* tf_binned_scatter returns hipErrorInvalidValue above this (the sort walks the keys in
                                 * LDS-sized ranges of 16384; ~1000^3 grids at 48 components stay below) */
/* Decompositions wider than 16 components are handled in 16-component groups (small per-workgroup LDS blocks); the groups
 * of a plane / line share its keys, the work-item table carries the group.  Number of keys for (grid, n_comp, T, LB), and
 * the number of (key, group) pairs per entry — what binned[] (an upper bound) and the work-item table are sized by: */
int tf_bin_nkeys(int model, const int grid[3], const int n_comp[3], int tile, int bucket, int share_groups);
int tf_bin_keys_per_entry(int model, const int n_comp[3]);
int tf_binned_scatter(const TfBinJob* job, tf_stream_t stream);
/* The sort stage (stage 1) of two jobs at once — three launches instead of six; b may be NULL. */
int tf_binned_sort_pair(const TfBinJob* a, const TfBinJob* b, tf_stream_t stream);

/* dst[j] = sum_r rep[r*stride + j], j < numel: folds the line-gradient replicas. */
int tf_reduce_replicas(const float* rep, int n_rep, int stride, int numel, float* dst, tf_stream_t stream);

/* Row gather / write-back of the data-parallel gradient exchange (parallel.py): `table` is the step's gradient buffer
 * viewed as rows of `w` floats (w % 4 == 0, 16-byte aligned), `idx` the n_rows row numbers that can be non-zero.
 * tf_gather_rows: packed[j] = table[idx[j]];  tf_scatter_rows: table[idx[j]] = packed[j] (idx holds no duplicates). */
int tf_gather_rows(const float* table, const int* idx, int n_rows, int w, float* packed, tf_stream_t stream);
int tf_scatter_rows(float* table, const int* idx, int n_rows, int w, const float* packed, tf_stream_t stream);

/* ---- regularisers of the training loop, TensorVMSplit ---------------------------------
 * loss[0] += w_ortho * vector_comp_diffs() + w_l1 * density_L1() + w_tv_density * TV_loss_density(TVLoss())
 *            + w_tv_app * TV_loss_app(TVLoss())            (train.py:340-371, tensoRF.py:175-205, loss.py:120-141)
 * and, with want_grad, the gradient of that sum times *scale (NULL = 1) is ADDED to density_grad / app_grad
 * (layouts of the factor tensors, n_rep ignored).  loss[1] += TV terms, loss[2] += L1 term, loss[3] += ortho term
 * (weighted).  `loss` is 4 device floats the caller zeroes.  Requires every n_comp % 4 == 0 and <= 64. */
typedef struct TfRegJob {
    TfFactors density, app;
    TfFactorGrads density_grad, app_grad;
    int grid[3];
    float w_ortho, w_l1, w_tv_density, w_tv_app;
    float* loss;
    const float* scale;
    int want_grad;
    const float* weights_dev;  /* NULL, or 4 DEVICE floats [ortho, l1, tv_density, tv_app] that multiply the host weights
                                * above (set those to 1): train.py:336-339 decays the TV weights every iteration, which a
                                * hipGraph capture would otherwise freeze */
} TfRegJob;
int tf_regularizers(const TfRegJob* job, tf_stream_t stream);

/* ---- optimizer step ---------------------------------------------------------------
 * Replaces `optimizer.step()` of train.py:376 for `torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`
 * (train.py:272-273; eps 1e-8, no weight decay, no amsgrad).  A segment is one parameter tensor walked in
 * storage order: p, g (its gradient, same layout), m / v (first / second moment), n elements, `group` selects
 * the learning rate lrs[group] (train.py uses two: lr_init for the factor tensors, lr_basis for the networks).
 * chunk_end[s] = number of TF_ADAM_CHUNK-element chunks in segments 0..s (one workgroup per chunk).
 * `lrs` and `step` are DEVICE pointers (float).
 *
 * Step counts are PER SEGMENT, like torch.optim.Adam's per-parameter state['step']: step[s] is the number of updates
 * segment s has COMPLETED, this launch is its update t = step[s] + 1, and the last workgroup to finish advances the
 * counts of the segments that were updated (`arrivals`: one zeroed device uint per job, re-armed by the kernel).
 * A segment is NOT updated — moments, parameters and its count stay as they are, exactly what torch.optim.Adam does
 * with a parameter whose .grad is None — when
 *   - its bit in `skip_mask` is set (the host knows the parameter has no gradient this step), or
 *   - it has a gate (`gate & 3`: 1 = live[0], 2 = live[1]) whose device word is zero and none of the regulariser
 *     flags selected by `gate >> 4` (bit 0 ortho, 1 L1, 2 TV-density, 3 TV-app of reg_active[4]) is non-zero.
 * Why: in the reference forward the density factors enter the autograd graph only `if ray_valid.any()`
 * (tensorBase.py:359-364) and the appearance factors, basis matrix and MLP only `if app_mask.any()` (:370-373); in a
 * step without such samples (the first iterations of a fresh field) their .grad stays None, Adam skips them, and their
 * bias correction later starts from t = 1.  live[0] / live[1] = number of density / shaded samples of the step
 * (TfLive, written by tf_composite_forward); the regulariser terms of train.py:340-371 give the factor tensors a
 * gradient regardless of the samples, hence the flags. */
#define TF_ADAM_MAX_SEG 32
#define TF_ADAM_CHUNK 8192
typedef struct TfAdamSeg {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    int group;
    int gate;                 /* 0: always updated; else (count index 1 | 2) | (regulariser bits << 4) */
} TfAdamSeg;
typedef struct TfAdamJob {
    int n_seg;
    int clear_grads;          /* 1: every non-zero gradient element is set to zero once it has been consumed (the g pointers
                               * are written through): the buffer is all zeros again when the launch ends, so the next
                               * backward can accumulate into it without a fill of its own (graph.GraphedTrainStep) */
    TfAdamSeg seg[TF_ADAM_MAX_SEG];
    int chunk_end[TF_ADAM_MAX_SEG];
    const float* lrs;
    float* step;              /* n_seg device floats: completed updates per segment (read, then advanced) */
    double beta1, beta2, eps;
    unsigned int* arrivals;   /* one zeroed device counter per job */
    unsigned int* touched;    /* NULL, or one word per workgroup (chunk of TF_ADAM_CHUNK elements), zero when the
                               * moments are created: bit set = that 256-float piece has had a non-zero gradient;
                               * pieces whose bit is clear have zero moments, which are then not read */
    const float* live;        /* NULL (no gates), or 3 device floats: density samples, shaded samples of this step, and the
                               * overflow flag of its right-sized lists (non-zero: NO segment is updated) */
    const float* reg_active;  /* NULL, or 4 device floats: non-zero = that regulariser term is part of the loss */
    unsigned int skip_mask;   /* bit s: segment s has no gradient this step */
    int pad_;
} TfAdamJob;
int tf_adam_step(const TfAdamJob* job, tf_stream_t stream);

/* Library identification: returns the gfx target string the kernels were compiled for. */
const char* tf_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* TENSORF_HIP_H */

/*
 * nb_hip.h — C ABI of libnb_hip.so, the MI355X (gfx950) implementation of the
 * Neural Body rendering hot path.
 *
 * Conventions:
 *   - every entry point returns 0 on success, a negative NB_E* code on failure;
 *     nb_last_error() returns a thread-local description of the last failure.
 *     Nothing throws across the ABI, nothing calls exit().
 *   - all pointers marked "dev" are DEVICE pointers owned by the caller (PyTorch-ROCm
 *     tensors: tensor.data_ptr()).  The library never allocates or frees user-visible
 *     memory, links no vendor BLAS and keeps no hidden state between calls.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     work is enqueued asynchronously, no entry point synchronises the device.
 *   - floats are fp32, indices int32 unless stated.
 *
 * Each entry point names the reference interface it replaces (paths are into
 * zju3dv/neuralbody, i.e. /root/reference).
 */
#ifndef NB_HIP_H
#define NB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_OK 0
#define NB_EINVAL (-1)  /* bad argument (null pointer, unsupported size) */
#define NB_ELAUNCH (-2) /* HIP launch / runtime error */
#define NB_ENODEV (-3)  /* no gfx950 device */

#define NB_ABI_VERSION 20

/* arithmetic of the decoder (nb_decode_points / nb_march `precision` argument) */
#define NB_PREC_F32 0   /* exact fp32 on v_mfma_f32_32x32x2_f32, trilinear gather of the dense volumes on the VALU: the reference's
                           own precision (the training forward, the activation tap, debugging) */
#define NB_PREC_F16F6 1 /* the default.  fc_0 FOLDED INTO THE VOLUME: trilinear interpolation and fc_0 are both linear with
                           nothing between them (latent_xyzc.py:62-72,99), so fc_0 . interp(V) = interp(fc_0 . V).
                           nb_fold_build stores U_l = fc_0[:, level l] . V_l per ACTIVE voxel (256 channels, fp16 head + fp16
                           remainder); the kernel fetches the rows of the voxels its 64 samples touch and contracts them with
                           the sparse trilinear-weight matrix Wt [voxel x sample] on the matrix pipe (three fp16 products per
                           K = 16 voxels, products exact, fp32 accumulate: fp32-level accuracy).  Needs nb_scene.fold.
                           fc_1, fc_2 and the colour head (feature_fc . latent_fc . view_fc folded into one layer): fp16 head x
                           fp16 head on v_mfma_f32_32x32x16_f16 + the two head x remainder cross terms on
                           v_mfma_scale_f32_32x32x64_f8f6f4 (weights fp4 e2m1 with a pack-time E8M0 scale per row and 32 K — fp6 e2m3
                           until ABI 19; the build flag NB_FP4_TERMS selects — activations bf6 e3m2 with a run-time E8M0 scale per
                           sample and 32 K), fp32 accumulate: ~2^-13 relative per cross term, RGB <= 4e-5 on every fixture.
                           Weight blocks whose elements span more than ~2^3 lose their small elements: see
                           nb_mlp_six_bit_stats_offset().  Rays whose LAST density it cannot sign: nb_march's ill_scratch. */

/* MLP geometry fixed by lib/networks/latent_xyzc.py:20-28 */
#define NB_FEAT_DIM 352 /* 32 + 64 + 128 + 128 interpolated channels */
#define NB_HID 256
#define NB_VIEW_HID 128
#define NB_N_LEVELS 4
#define NB_TAP_WIDTH 1600

const char *nb_last_error(void);
int nb_abi_version(void);
/* returns the number of visible HIP devices (>=0) or a negative error */
int nb_device_count(void);

/* ---------------------------------------------------------------------------------
 * Scene description shared by the decode / march entry points.
 * Mirrors `sp_input` built at lib/networks/renderer/if_clight_renderer.py:29-52 plus
 * the four feature volumes returned by Network.encode_sparse_voxels
 * (lib/networks/latent_xyzc.py:30-39).  Volumes are CHANNELS-LAST: vol[l] is a dense
 * [D_l, H_l, W_l, C_l] fp32 array (C_l = 32, 64, 128, 128), zeros at inactive voxels.
 * ------------------------------------------------------------------------------- */
/* fc_0 folded into the four latent volumes (NB_PREC_F16F6), built by nb_fold_build from the SAME volumes and fc_0 weight:
 * row r of level l = fc_0.weight[:, channels of level l] . V_l[voxel of row r]  (256 outputs) as 256 fp16 heads followed by 256
 * fp16 remainders (1 KiB); grid[l] = index grid of the level ([D_l,H_l,W_l] int32: row id inside the level or -1 = no row). */
typedef struct nb_fold {
    const uint16_t *urows;            /* dev [(rows of all levels) + 1][512]; the extra last row is all zero */
    const int32_t *grid[NB_N_LEVELS]; /* dev */
    int32_t row_base[NB_N_LEVELS];    /* first row of each level inside urows */
    int32_t zero_row;                 /* index of the all-zero row (inactive voxels, padding) */
} nb_fold;

typedef struct nb_scene {
    const float *vol[NB_N_LEVELS]; /* dev */
    int32_t vol_dhw[NB_N_LEVELS][3];
    /* dev, NB_POSE_FLOATS floats: R[9] row-major (canonical = (p_world - Th) @ R, latent_xyzc.py:41-47) | Th[3] |
     * bounds_min[3] (SMPL-space AABB minimum, xyz order = sp_input['bounds'][0]).  The per-frame tensors of sp_input
     * are read by the kernels themselves: they never visit the host, so nothing about a frame is cached host-side. */
    const float *pose;
    float voxel_size[3]; /* cfg.voxel_size, dhw order (latent_xyzc.py:54) */
    int32_t out_sh[3];   /* full-resolution grid D,H,W (sp_input['out_sh']) */
    const nb_fold *fold; /* HOST pointer or NULL; required by NB_PREC_F16F6 (which does not read vol[]) */
} nb_scene;
#define NB_POSE_FLOATS 15

/* ---------------------------------------------------------------------------------
 * Packed decoder weights.  nb_mlp_pack_size() floats; produced by nb_mlp_pack() from
 * the reference's parameter tensors (Conv1d(k=1) weights [out,in,1] viewed as [out,in]).
 * The blob holds both formats (fp32 fragments for NB_PREC_F32, the fp16 / fp4 fragment stream of NB_PREC_F16F6).
 * The packing re-orders every layer into MFMA A-operand fragment order and merges
 * feature_fc with the first 256 columns of latent_fc (no activation sits between them,
 * latent_xyzc.py:106-111).  nb_mlp_latent_bias() folds the per-frame latent code
 * (latent_xyzc.py:108-111) into that merged layer's bias; call it whenever
 * latent_index or the parameters change.
 * ------------------------------------------------------------------------------- */
int64_t nb_mlp_pack_size(void);        /* floats in the packed blob */
int64_t nb_mlp_latent_bias_size(void); /* floats in the per-frame bias block (384: see nb_mlp_latent_bias) */
/* Float offset inside the packed blob of 6 int32 counters written with the NB_PACK_F16F6 section: for each of the three
 * layers that kernel runs with narrow cross terms (fc_1, fc_2, the folded feature_fc / latent_fc / view_fc layer) {small, nonzero}
 * = how many non-zero weights lie below 1/8 of the maximum of their (row, 32 K) block — where fp6 e2m3 keeps fewer than 3 and fp4 e2m1 no
 * significant bits — and how many are non-zero at all.  small / nonzero is ~0.2 for normally distributed weights; a caller that
 * cannot rule out weight blocks with a wide dynamic range (> ~2^5) uses it to fall back to NB_PREC_F32 (the Python Network does,
 * precision "auto"). */
int64_t nb_mlp_six_bit_stats_offset(void);

typedef struct nb_mlp_params { /* all dev, row-major [out,in] / [out] */
    const float *fc0_w, *fc0_b;         /* [256,352] */
    const float *fc1_w, *fc1_b;         /* [256,256] */
    const float *fc2_w, *fc2_b;         /* [256,256] */
    const float *alpha_w, *alpha_b;     /* [1,256]   */
    const float *feature_w, *feature_b; /* [256,256] */
    const float *latent_w, *latent_b;   /* [256,384] */
    const float *view_w, *view_b;       /* [128,346] */
    const float *rgb_w, *rgb_b;         /* [3,128]   */
} nb_mlp_params;

/* replaces: module parameter access in Network.calculate_density_color
 * (lib/networks/latent_xyzc.py:99-121).  `packed` dev, nb_mlp_pack_size() floats. */
int nb_mlp_pack(const nb_mlp_params *p, float *packed, void *stream);
/* The same, writing only the sections a caller is going to use (a training step changes the weights every iteration and
 * decodes with NB_PREC_F32 only).  The fp32 section (NB_PACK_F32) is always written. */
#define NB_PACK_F32 1
#define NB_PACK_F16F6 2 /* the weight stream of NB_PREC_F16F6 (fc_1, fc_2, the folded colour head; fc_0 lives in nb_fold) */
#define NB_PACK_ALL 3
int nb_mlp_pack_sections(const nb_mlp_params *p, float *packed, int sections, void *stream);
/* latent_row: dev pointer to latent.weight[latent_index] (128 floats);
 * out: dev, nb_mlp_latent_bias_size() floats = [256: bias of the merged feature_fc / latent_fc layer with the latent code
 * folded in | 128: bias of view_fc with that layer folded in as well (feature_fc, latent_fc and view_fc have no activation
 * between them, latent_xyzc.py:105-119; NB_PREC_F16F6 runs them as one layer)], MFMA fragment order. */
int nb_mlp_latent_bias(const nb_mlp_params *p, const float *latent_row, float *out, void *stream);

/* ---------------------------------------------------------------------------------
 * nb_fold_build — the encoder-side half of NB_PREC_F16F6: replaces the first decoder layer's weight access in
 * Network.calculate_density_color (lib/networks/latent_xyzc.py:99, `self.fc_0(features)`) together with the four
 * F.grid_sample calls feeding it (:62-72), by pre-multiplying every ACTIVE voxel of the four volumes with fc_0.
 *   vol[l] dev channels-last [D_l,H_l,W_l,C_l] fp32 (nb_scene.vol); rows_lin[l] dev [n_rows_max[l]] int32: linear voxel index
 *   of each active row — or rows_lin[l] NULL: vol[l] then holds the level's ACTIVE ROWS themselves, compact [n_rows_max[l], C_l]
 *   (what the encoder produces before `.dense()`, latent_xyzc.py:188-201: no dense volume need exist for this arithmetic);
 *   n_rows[l] dev [1] int32 (device-side count, <= n_rows_max[l]); fc0_w dev [256,352] fp32 row-major;
 *   urows dev [(sum of n_rows_max) + 1][512] uint16: level l's rows start at row sum_{k<l} n_rows_max[k]; the last row is
 *   zero-filled by the call (sum of n_rows_max < 2^21).  Exact fp32 products (v_mfma_f32_32x32x2_f32), then head = fp16(u),
 *   remainder = fp16(u - head).  n_saturated dev [1] int32 or NULL, ZEROED BY THE CALLER (the call adds to it; the
 *   encoder's one zero fill covers it): how many products were beyond the fp16 range (+-65504: clamped) or not finite (a NaN
 *   product stays NaN in the planes) — the planes then do not carry fc_0 . V, use
 *   NB_PREC_F32 for such weights / volumes.
 * nb_sparsify — active set of a DENSE volume that did not come with one (volumes handed to Network.calculate_density_color by a
 * caller other than encode_sparse_voxels): a voxel is active iff any channel is non-zero.  grid dev [D*H*W] int32 out (row id
 * or -1), rows_lin dev [n_rows_max] out (linear-voxel order), n_rows dev [1] out (clamped to n_rows_max; rows beyond it are
 * dropped: size n_rows_max = D*H*W to be safe); scratch dev nb_scan_scratch_size(D*H*W) bytes.
 * ------------------------------------------------------------------------------- */
int nb_fold_build(const float *const vol[NB_N_LEVELS], const int32_t *const rows_lin[NB_N_LEVELS],
                  const int32_t *const n_rows[NB_N_LEVELS], const int32_t n_rows_max[NB_N_LEVELS], const float *fc0_w,
                  uint16_t *urows, int32_t *n_saturated, void *stream);
int nb_sparsify(const float *vol, const int32_t dhw[3], int32_t c, int32_t *grid, int32_t *rows_lin, int32_t *n_rows,
                int32_t n_rows_max, void *scratch, void *stream);

/* ---------------------------------------------------------------------------------
 * nb_decode_points — replaces Network.calculate_density_color
 * (lib/networks/latent_xyzc.py:91-126) and, with density_only != 0,
 * Network.calculate_density (:74-89).
 *   wpts    dev [n,3] world-space points; viewdir dev [n,3] (ignored when density_only)
 *   raw_out dev [n,4] (rgb logits, sigma)   or [n,1] sigma when density_only
 *   dbg     dev or NULL (NB_PREC_F32 only): activation tap, NB_TAP_WIDTH floats per point:
 *           [F 352 | h1 256 | h2 256 | h3 256 | G 256 (latent_fc output) | V 128 | PE 90 | pad]
 *           (post-ReLU where the layer has one) — what the backward pass consumes; also used by tests
 * ------------------------------------------------------------------------------- */
int nb_decode_points(const nb_scene *scene, const float *packed, const float *latent_bias,
                     const float *wpts, const float *viewdir, int64_t n, int density_only,
                     float *raw_out, float *dbg, int precision, void *stream);

/* ---------------------------------------------------------------------------------
 * Optional sample culling of nb_march — replaces prepare_inside_pts of the mask-culled renderers
 * (lib/networks/renderer/if_clight_renderer_mmsk.py:12-45, if_clight_renderer_msk.py:12-49): a sample
 * is decoded only if it projects inside every one of n_views silhouettes; culled samples get raw = 0
 * (if_clight_renderer_mmsk.py:54-59).  pre_affine != 0 (the _msk variant) first maps the point
 * world -> SMPL space of the rendered pose (scene R, Th) -> world of the snapshot frame (R0, Th0).
 * ------------------------------------------------------------------------------- */
#define NB_SLOT_DEAD (-2147483647 - 1) /* nb_march ray_order: empty slot */
#define NB_MAX_CULL_VIEWS 64 /* the reference loops over batch['Ks'].size(1) training views (4 or 6 in the shipped configs) */
#define NB_CULL_CAM_FLOATS 21
typedef struct nb_cull {
    int32_t n_views;    /* 1..NB_MAX_CULL_VIEWS */
    int32_t H, W;       /* mask size = int(cfg.H * cfg.ratio), int(cfg.W * cfg.ratio) */
    int32_t pre_affine;
    const uint8_t *msk; /* dev [n_views,H,W] uint8, non-zero = inside (batch['msks'][0] / batch['msk']) */
    const float *cam;   /* dev [n_views, NB_CULL_CAM_FLOATS]: row-major 3x4 [R|T] (batch['RT']) | K 3x3 (batch['Ks'] / batch['K']) */
    const float *snap;  /* dev 12 floats: batch['R0_snap'] (9) | batch['Th0_snap'] (3); NULL unless pre_affine */
} nb_cull;

/* ---------------------------------------------------------------------------------
 * nb_march — the fused per-ray path (both precisions serve sample culling and the raw output): replaces Renderer.get_pixel_value
 * (lib/networks/renderer/if_clight_renderer.py:62-92), i.e. get_sampling_points (:11-27),
 * get_density_color (:54-60), Network.calculate_density_color and raw2outputs
 * (lib/networks/renderer/nerf_net_utils.py:6-51, raw_noise_std = 0), for ALL rays of a
 * batch element in one launch (the reference's 2048-ray chunk loop :107-118 disappears).
 *   ray_o, ray_d dev [n_rays,3]; near, far dev [n_rays]
 *   t_vals  dev [n_samples]  = torch.linspace(0,1,n_samples)   (if_clight_renderer.py:13)
 *   t_rand  dev [n_rays,n_samples] in [0,1) or NULL            (stratified jitter, :16-23)
 *   ray_order dev [n_slots] int32 or NULL (n_slots ignored, slot i = ray i): WHICH rays march together.  64 consecutive slots
 *           share a workgroup (whose voxel list is the union of what its samples touch: the caller groups neighbouring
 *           pixels, e.g. 8 x 8 tiles).  Entry v >= 0: the slot marches ray v and stores its results at the ray's own index;
 *           v = -(r + 1) in (NB_SLOT_DEAD, 0): a padding slot that marches ray r's data and stores nothing (keeps a partially
 *           filled tile's group together); NB_SLOT_DEAD: an empty slot — a group of 64 (aligned) slots is either free of
 *           them or consists of them (it is skipped).  Every ray must appear exactly once as v >= 0.  n_slots % 64 == 0.
 *   cull    HOST pointer to an nb_cull (whose msk / cam / snap members are DEVICE pointers) or NULL (no culling)
 *   outputs dev: rgb_map [n_rays,3], disp_map/acc_map/depth_map [n_rays],
 *           weights [n_rays,n_samples]; raw (optional, may be NULL) [n_rays,n_samples,4]
 *   ill_scratch dev, ill_scratch_bytes = NB_ILL_SCRATCH_BYTES(cap) for a list of `cap` >= 1 rays (16-byte aligned), or NULL / 0;
 *           NB_PREC_F16F6 only (NB_PREC_F32 ignores it).  The reference
 *           gives a ray's LAST sample the interval 1e10 (nerf_net_utils.py:28), so that sample's alpha is a step function of the SIGN
 *           of its density; NB_PREC_F16F6's density error (~3e-4 absolute on the bench scene) can put a ray whose last density is
 *           that close to zero on the other side of the step (~4e-6 of the rays, each off by up to its remaining transmittance).  With
 *           the scratch the march LISTS every ray whose last density is within NB_ILL_SIGMA of zero while its transmittance in front
 *           of that sample exceeds NB_ILL_T_MIN (state in front of the sample + the sample's colour logits, <= cap rays), and a
 *           second small kernel on the same stream recomputes that one density per listed ray at fp32 level (fc_0 through the folded
 *           planes' head + remainder pairs, fc_1 / fc_2 / alpha_fc from the fp32 section of `packed`, fp64 accumulation), composites
 *           the last sample again and rewrites the ray's rgb / disp / acc / depth, weights[n_samples - 1] and raw[n_samples - 1][3].
 *           No host read-back.  Afterwards the scratch starts with int32 {rays listed (may exceed cap: the excess keeps the
 *           march's result), rays whose step changed side}.  The call zeroes that header itself.  NULL: the march's result as is.
 *           The bench view lists ~50 of its 262 144 rays (~20 us); a cap of a few thousand covers any trained scene, whose empty
 *           space has a strongly negative density — only a decoder whose density is ~0 over whole regions lists rays by the
 *           thousand (12 533 listed rays: +0.5 ms), which is what the cap bounds.
 * ------------------------------------------------------------------------------- */
#define NB_ILL_SIGMA 4e-3f  /* > 10 x the measured density error of NB_PREC_F16F6 on the bench scene (alpha_fc weights x 50) */
#define NB_ILL_T_MIN 1e-6f  /* a last sample in front of which less transmittance is left cannot move the ray by more */
#define NB_ILL_HEADER_FLOATS 16
#define NB_ILL_RECORD_FLOATS 16
#define NB_ILL_FIXUP_BLOCKS 1024
#define NB_ILL_SCRATCH_BYTES(cap) (4 * ((int64_t)NB_ILL_HEADER_FLOATS + (int64_t)(cap) * NB_ILL_RECORD_FLOATS))
int nb_march(const nb_scene *scene, const float *packed, const float *latent_bias,
             const float *ray_o, const float *ray_d, const float *near, const float *far,
             int64_t n_rays, int32_t n_samples, const float *t_vals, const float *t_rand,
             const int32_t *ray_order, int64_t n_slots, const nb_cull *cull, int white_bkgd, float *rgb_map, float *disp_map, float *acc_map, float *weights,
             float *depth_map, float *raw, void *ill_scratch, int64_t ill_scratch_bytes, int precision, void *stream);

/* ---------------------------------------------------------------------------------
 * nb_composite — raw2outputs alone (lib/networks/renderer/nerf_net_utils.py:6-51) for
 * callers that decode points themselves (the _mmsk/_msk renderers).
 *   raw dev [n_rays,n_samples,4]; z_vals dev [n_rays,n_samples]; ray_d dev [n_rays,3]
 * ------------------------------------------------------------------------------- */
int nb_composite(const float *raw, const float *z_vals, const float *ray_d, int64_t n_rays,
                 int32_t n_samples, int white_bkgd, float *rgb_map, float *disp_map,
                 float *acc_map, float *weights, float *depth_map, void *stream);

/* ---------------------------------------------------------------------------------
 * Backward pass (training step: lib/train/trainers/if_nerf_clight.py:18-36 runs Renderer.render
 * under autograd; the reference's backward is PyTorch autograd through the same modules).
 * ------------------------------------------------------------------------------- */

/* d raw [n_rays,n_samples,4] from d rgb_map [n_rays,3] (and optionally d acc_map / d depth_map
 * [n_rays], may be NULL): backward of raw2outputs (nerf_net_utils.py:19-49, raw_noise_std 0). */
int nb_composite_bwd(const float *raw, const float *z_vals, const float *ray_d, int64_t n_rays,
                     int32_t n_samples, int white_bkgd, const float *d_rgb_map, const float *d_acc_map,
                     const float *d_depth_map, float *d_raw, void *stream);

/* Row-major fp32 GEMM C[m,n] = alpha * op(A) op(B) + beta * C on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact
 * fp32): the backward of the Conv1d(k=1) layers (latent_xyzc.py:99-121).  lda/ldb/ldc are row strides.  op(A) = A^T
 * (trans_a) is the weight-gradient form dW = dY^T X (split over the long row dimension, fp32 atomics; for M, N >= 32, K >= 1024
 * and 16-byte aligned rows it runs on the 16-bit matrix pipe with both operands as bf16 head + remainder pairs, ~2^-16
 * relative, else exact fp32; op(B) = B only:
 * trans_a && trans_b is refused with NB_EINVAL, no caller of the path needs it).  Summation order: the trans_a form and the
 * colsum epilogue of nb_gemm_fused accumulate with fp32 atomics, so weight / bias gradients are reproducible to rounding
 * (~1e-7 relative), not bit for bit, from run to run — like the reference's own cuBLAS / atomics-based backward.  k == 0 is
 * an empty product: C = beta C (and the epilogue). */
int nb_sgemm(int trans_a, int trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float *a, int32_t lda,
             const float *b, int32_t ldb, float beta, float *c, int32_t ldc, void *stream);
/* (The non-transposed-A form with op(B) = B, >= 1024 rows, K a multiple of 32 and 16-byte aligned operands — the dX = dY . W
 * products — also runs on the 16-bit matrix pipe with bf16 pairs.)
 * The same with the epilogues of the backward chain fused (trans_a = 0 only):
 *   mask_y (dev [m, >= n], row stride ldy, or NULL): C[r,c] = 0 where mask_y[r,c] <= 0 — the ReLU that followed the
 *            layer whose input gradient C is (mask_y = that layer's post-activation output);
 *   colsum (dev [n] or NULL): colsum[c] += sum_r C[r,c] after the mask — the bias gradient of that layer. */
int nb_gemm_fused(int trans_a, int trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float *a, int32_t lda,
                  const float *b, int32_t ldb, float beta, float *c, int32_t ldc, const float *mask_y, int32_t ldy,
                  float *colsum, void *stream);

/* dy[i] = y[i] > 0 ? dy[i] : 0 (ReLU backward on the post-activation value), in place. */
int nb_relu_bwd(float *dy, const float *y, int64_t n, void *stream);
/* out[c] += sum_r x[r*ld + c]  (bias gradients); out must be initialised by the caller. */
int nb_colsum(const float *x, int64_t n_rows, int32_t n_cols, int32_t ld, float *out, void *stream);

/* Backward of the 4-level trilinear lookup (F.grid_sample, latent_xyzc.py:62-72) restricted to the
 * ACTIVE voxels: d_feat dev [n,352] -> drows[l] dev [n_rows_l, C_l] (+=, atomics; zero them first),
 * grids[l] dev = index grid of level l (row id or -1).  scene->vol[] is not read.
 * run_length >= 1: consecutive points [k run_length, (k+1) run_length) are the samples of ONE ray in depth order (the training
 * step: N_samples); contributions to the same base voxel are summed along such a run before they are added atomically (the
 * result is the same sum in another order).  1 = every point on its own. */
int nb_trilinear_bwd(const nb_scene *scene, const int32_t *const grids[4], float *const drows[4],
                     const float *wpts, const float *d_feat, int64_t n, int32_t run_length, void *stream);

/* ---------------------------------------------------------------------------------
 * Structured-latent-code encoder — replaces Network.encode_sparse_voxels
 * (lib/networks/latent_xyzc.py:30-39) + SparseConvNet.forward (:184-205) and the spconv
 * v1.2.1 calls inside (SubMConv3d / SparseConv3d / BatchNorm1d / ReLU / .dense()).
 * A sparse tensor is (rows [n,C] fp32, coords [n] linear voxel index, index grid
 * [D,H,W] int32 holding the row id or -1).
 * ------------------------------------------------------------------------------- */

/* Voxelise the SMPL vertices: dedup (last vertex index wins, spconv_standin rule),
 * build the index grid and compact row list.
 *   coord dev [n_verts,3] (d,h,w) int32; grid dev [D*H*W] int32 (overwritten; filled with -1 by the call unless flags has
 *   NB_GRID_PREFILLED);
 *   rows_vert dev [n_verts] int32 out: vertex id feeding each row; rows_lin dev [n_verts]
 *   int32 out: linear voxel index of each row; n_rows dev [1] int32 out.
 *   scratch dev: at least nb_scan_scratch_size(n_verts) bytes. */
int64_t nb_scan_scratch_size(int64_t n);
#define NB_GRID_PREFILLED 1 /* flags of nb_enc_voxelize / nb_enc_downsample_index: the caller filled the index grid with -1 (one fill
                               for the grids of all five levels of an encoder pass instead of one memset each) */
int nb_enc_voxelize(const int32_t *coord, int32_t n_verts, const int32_t dhw[3], int32_t *grid,
                    int32_t *rows_vert, int32_t *rows_lin, int32_t *n_rows, void *scratch, int32_t flags,
                    void *stream);

/* Output active set of SparseConv3d(k=3, s=2, p=1) (latent_xyzc.py:265-274): every output
 * site with >=1 active input in its 3^3 receptive field, rows numbered in linear-voxel
 * order.  in_lin dev [n_in_max] with *n_in valid (device count); out grid dev [Do*Ho*Wo]. */
int nb_enc_downsample_index(const int32_t *in_lin, const int32_t *n_in, int32_t n_in_max,
                            const int32_t in_dhw[3], const int32_t out_dhw[3], int32_t *out_grid,
                            int32_t *out_lin, int32_t *n_out, int32_t n_out_max, void *scratch, int32_t flags,
                            void *stream);
/* The index sets of n_levels (<= NB_DOWN_LEVELS_MAX) SUCCESSIVE strided levels below a voxelised level in three launches instead of
 * three per level: level l's out_dhw is the l-fold floor((d - 1) / 2) + 1 of in_dhw; out_grid[l], out_lin[l], n_out[l], n_out_max[l] (host
 * arrays of device pointers / capacities) as in nb_enc_downsample_index.  A cell c of level l is active iff an active voxel p of the
 * base level lies within [2^l c - (2^l - 1), 2^l c + (2^l - 1)] in every coordinate: the per-level rule composed, and the same
 * cells, rows and order as n_levels chained nb_enc_downsample_index calls as long as no capacity clamps (the capacities of
 * ops.down_capacity are upper bounds).  scratch: nb_scan_scratch_size(max(cells of level 1, 64)) bytes. */
#define NB_DOWN_LEVELS_MAX 4
int nb_enc_downsample_index_all(const int32_t *in_lin, const int32_t *n_in, int32_t n_in_max, const int32_t in_dhw[3],
                                int32_t n_levels, int32_t *const out_grid[], int32_t *const out_lin[], int32_t *const n_out[],
                                const int32_t n_out_max[], void *scratch, int32_t flags, void *stream);

/* One sparse 3x3x3 convolution (stride 1 submanifold or stride 2) without bias:
 *   out[r, :] = sum_o W[o] . in[nbr(r, o), :]   over ACTIVE neighbours
 * and per-channel sum / sum of squares of the result rows (fp64) for BatchNorm.
 *   weight dev [3,3,3,Cin,Cout] (spconv 1.x layout); in_rows dev [n_in,Cin];
 *   in_grid dev index grid of the INPUT tensor; out_lin dev [n_out_max] linear voxel index
 *   of each OUTPUT row in the OUTPUT grid; n_out dev [1]; stats dev [2*Cout] fp64 (zeroed
 *   by the call unless flags has NB_CONV_STATS_ZEROED: the caller cleared it, e.g. one memset for the statistics of all
 *   17 layers instead of one launch per layer). */
#define NB_CONV_STATS_ZEROED 1
#define NB_CONV_BF16 2 /* nb_enc_conv16 only: the split operands are bf16 pairs (nb_enc_conv_pack16 mode 1, nb_enc_bn_relu_bwd dx_split) */
int nb_enc_conv(const float *in_rows, const int32_t *in_grid, const int32_t in_dhw[3],
                const int32_t *out_lin, const int32_t *n_out, int32_t n_out_max,
                const int32_t out_dhw[3], int32_t stride, const float *weight, int32_t cin,
                int32_t cout, float *out_rows, double *stats, int32_t flags, void *stream);

/* BatchNorm1d(eps=1e-3) over active rows + ReLU, in place (latent_xyzc.py:208-274).
 * training != 0: normalise with the batch statistics in `stats` (biased variance), write
 * [mean | biased var | n_rows] (2*C+1 floats) to batch_stats (dev, may be NULL when momentum < 0)
 * and, when momentum >= 0, update running_mean / running_var in place like nn.BatchNorm1d
 * (unbiased variance, latent_xyzc.py:215 momentum 0.01);
 * training == 0: use running_mean / running_var.
 * dense (dev or NULL): channels-last [D,H,W,C] volume receiving the rows (.dense(),
 * latent_xyzc.py:189-201); it must have been zero-filled by the caller.
 * rows_out (dev or NULL): when given, the activated rows go there and `rows` keeps the raw
 * convolution output (needed by nb_enc_bn_relu_bwd). */
int nb_enc_bn_relu(float *rows, const int32_t *n_rows, int32_t n_rows_max, int32_t c,
                   const double *stats, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, int training, float eps, float momentum,
                   float *batch_stats, const int32_t *rows_lin, float *dense, float *rows_out, void *stream);

/* ---- the same convolution on the 16-bit matrix pipe (inference: no backward record) ----
 * Operands as fp16 head + fp16 remainder, three products per K chunk on v_mfma_f32_32x32x16_f16 with fp32 accumulation
 * (~2^-21 relative per product), 1.3-1.9x faster than nb_enc_conv.  cin in {32, 64, 128}, cout in {32, 64, 128}.
 *   nb_enc_conv_pack16: weight dev [3,3,3,Cin,Cout] fp32 -> packed dev, 27*Cin*Cout*2 uint16 (MFMA B-fragment order).
 *     mode 0: the forward convolution, fp16 pairs.  mode 1: the BACKWARD-INPUT convolution of a layer as a
 *     convolution of its own (d in[q] = sum_o d out[q + o - 1] . W[26 - o]^T: mirrored offsets, transposed slabs), bf16 pairs
 *     (gradients span more binades than an un-scaled fp16 head holds): `cin`, `cout` are those of the packed convolution =
 *     the layer's Cout, Cin; `weight` is the layer's weight [3,3,3,cout,cin].  Run it with nb_enc_conv16(flags = NB_CONV_BF16)
 *     on the bf16 planes nb_enc_bn_relu_bwd writes (dx_split), in_grid = the layer's output grid, out_lin = its input rows.
 *   nb_enc_bn_relu_split: nb_enc_bn_relu whose activated rows leave as TWO fp16 planes in rows_split (dev, 2*n_rows_max*c
 *     uint16 = the bytes of an fp32 [n_rows_max, c] matrix: heads, then remainders); `rows` (the raw convolution output) is
 *     only read; dense as in nb_enc_bn_relu; rows_out (dev or NULL): the activated rows in fp32 as well (the training
 *     forward keeps them for nb_enc_bn_relu_bwd / nb_enc_conv_bwd_weight while the next convolution reads the planes)
 *   nb_enc_conv16: in_split = such a pair of planes with in_rows_cap rows each; everything else as nb_enc_conv, and
 *     stride = -2: the TRANSPOSED gather of a stride-2 layer (its backward-input product with a mode-1 pack): output row p (a row
 *     of the layer's input level) takes, under offset k, the row of voxel (p - 1 + k) / 2 of in_grid (the layer's output grid)
 *     where that division is exact in all three coordinates.  Channel pairs: (32,32) (32,64) (64,32) (64,64) (64,128) (128,64)
 *     (128,128) */
int nb_enc_conv_pack16(const float *weight, int32_t cin, int32_t cout, uint16_t *packed, int32_t mode, void *stream);
/* up to NB_PACK_BATCH_MAX nb_enc_conv_pack16 jobs in ONE launch (host arrays of device pointers and sizes): a training step re-packs
 * every >= 32-channel convolution after each optimiser step, forward (mode 0) and backward-input (mode 1) forms */
#define NB_PACK_BATCH_MAX 32
int nb_enc_conv_pack16_batch(int32_t n_jobs, const float *const weight[], const int32_t cin[], const int32_t cout[],
                             uint16_t *const packed[], const int32_t mode[], void *stream);
int nb_enc_bn_relu_split(const float *rows, const int32_t *n_rows, int32_t n_rows_max, int32_t c,
                         const double *stats, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, int training, float eps, float momentum,
                         float *batch_stats, const int32_t *rows_lin, float *dense, uint16_t *rows_split, float *rows_out,
                         void *stream);
int nb_enc_conv16(const uint16_t *in_split, int32_t in_rows_cap, const int32_t *in_grid, const int32_t in_dhw[3],
                  const int32_t *out_lin, const int32_t *n_out, int32_t n_out_max, const int32_t out_dhw[3],
                  int32_t stride, const uint16_t *wpacked, int32_t cin, int32_t cout, float *out_rows, double *stats,
                  int32_t flags, void *stream);

/* ---- encoder backward (training).  Shapes as in the forward calls above. ---- */

/* BatchNorm1d (batch statistics) + ReLU backward.  dy, y (activated rows), x (raw conv output) dev
 * [n_rows, c]; batch_stats dev [2c+1] as written by nb_enc_bn_relu; sums dev [2c] fp64 scratch.
 * Outputs: dx dev [n_rows, c] (may alias dy), dgamma / dbeta dev [c]; dx_split (dev or NULL): dx once more as two bf16
 * planes [2, n_rows_max, c] (heads | remainders) for the backward-input convolution on the matrix pipe. */
#define NB_BWD_RULEBOOK_READY 2 /* nb_enc_conv_bwd_weight: `rulebook` already holds nbr(r, o) of this (in_grid, out_lin, stride) — written by
                                  an earlier call for a layer of the same level (its submanifold layers share one table) */
#define NB_BWD_ZEROED 1 /* flags of nb_enc_bn_relu_bwd / nb_enc_conv_bwd_weight: the caller cleared `sums` / `dweight` (one zero fill
                           for the accumulators of a whole backward pass instead of one memset per layer) */
int nb_enc_bn_relu_bwd(const float *dy, const float *y, const float *x, const int32_t *n_rows,
                       int32_t n_rows_max, int32_t c, const float *batch_stats, float eps, const float *gamma,
                       double *sums, float *dx, float *dgamma, float *dbeta, uint16_t *dx_split, int32_t flags, void *stream);

/* Gradient of the conv INPUT rows: din[q] = sum_o dx[r(q,o)] @ W[o]^T, where r(q,o) is the output row that
 * read input voxel q under kernel offset o.  out_grid = index grid of the OUTPUT tensor, in_lin = linear voxel
 * index of each INPUT row. */
int nb_enc_conv_bwd_input(const float *dx, const int32_t *out_grid, const int32_t out_dhw[3], const int32_t *in_lin,
                          const int32_t *n_in, int32_t n_in_max, const int32_t in_dhw[3], int32_t stride,
                          const float *weight, int32_t cin, int32_t cout, float *din, void *stream);

/* Gradient of the conv weight [3,3,3,Cin,Cout] (zeroed by the call unless flags has NB_BWD_ZEROED): dW[o] = sum_r in[nbr(r,o)]^T (x) dx[r].
 * rulebook: dev int32 scratch [n_out_max * 27] (receives nbr(r,o), -1 = no active input voxel; only read with NB_BWD_RULEBOOK_READY).
 * dx_split (dev or NULL): dx as bf16 head / remainder planes [2, n_out_max, Cout] (nb_enc_bn_relu_bwd writes them); when given
 * and Cin >= 32 the product runs on the 16-bit matrix pipe with both operands as bf16 pairs (three products, fp32 accumulate,
 * ~2^-16 relative) instead of the exact-fp32 MFMA kernel. */
int nb_enc_conv_bwd_weight(const float *in_rows, const int32_t *in_grid, const int32_t in_dhw[3],
                           const int32_t *out_lin, const int32_t *n_out, int32_t n_out_max, const int32_t out_dhw[3],
                           int32_t stride, const float *dx, const uint16_t *dx_split, int32_t cin, int32_t cout,
                           float *dweight, int32_t *rulebook, int32_t flags, void *stream);

/* Embedding-lookup backward: dcodes[rows_vert[r], :] = drows[r, :] (dcodes zeroed by the caller). */
int nb_enc_scatter_codes_bwd(const float *drows, const int32_t *rows_vert, const int32_t *n_rows, int32_t n_rows_max,
                             int32_t c, float *dcodes, void *stream);

/* Embedding lookup of the per-vertex codes (latent_xyzc.py:33-34): rows[r,:] = c[rows_vert[r],:] */
int nb_enc_gather_codes(const float *codes, const int32_t *rows_vert, const int32_t *n_rows,
                        int32_t n_rows_max, int32_t c, float *rows, void *stream);

/* ---------------------------------------------------------------------------------
 * nb_raygen — replaces lib/utils/render_utils.py:120-137 (image_rays), i.e. get_rays
 * (lib/utils/if_nerf/if_nerf_data_utils.py:8-21) + get_near_far (:54-69) + compaction by
 * mask_at_box, on device.  K, R (row-major 3x3) and T (3) are HOST doubles.
 *   bounds: host float[6] world-space AABB (min xyz, max xyz)
 *   outputs dev: ray_o, ray_d [H*W,3]; near, far [H*W] (first *n_rays rows valid, pixel
 *   order preserved); mask_at_box [H*W] uint8; n_rays dev [1] int32.
 *   scratch dev: nb_scan_scratch_size(H*W) bytes. */
int nb_raygen(int32_t H, int32_t W, const double K[9], const double R[9], const double T[3],
              const float bounds[6], float *ray_o, float *ray_d, float *near, float *far,
              uint8_t *mask_at_box, int32_t *n_rays, void *scratch, void *stream);

/* ---------------------------------------------------------------------------------
 * nb_image_assemble — replaces the image re-assembly of the demo visualizer
 * (lib/visualizers/if_nerf_demo.py:15-30; same scatter in lib/evaluators/if_nerf.py:59-68):
 *   img = white_bkgd ? 1 : 0;  img[mask_at_box] = rgb_map;  (optionally) img = img[..., ::-1];  img *= scale
 *   depth = 0;  depth[mask_at_box] = depth_map
 * on device, so a view leaves the GPU (or enters the all-gather) as a finished image.
 *   mask_at_box dev [n_pixels] uint8; rgb_map dev [n_rays,3], depth_map dev [n_rays] or NULL — the
 *   compacted per-ray outputs in pixel order (what nb_raygen + nb_march produce); img dev [n_pixels,3];
 *   depth dev [n_pixels] or NULL (iff depth_map is NULL); scratch dev nb_scan_scratch_size(n_pixels) bytes.
 *   Pixels whose compacted index is >= n_rays keep the background. */
int nb_image_assemble(const uint8_t *mask_at_box, int64_t n_pixels, const float *rgb_map, const float *depth_map,
                      int64_t n_rays, int white_bkgd, int bgr, float scale, float *img, float *depth, void *scratch,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NB_HIP_H */

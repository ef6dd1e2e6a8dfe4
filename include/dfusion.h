/*
 * dfusion.h -- C-ABI of the MI355X (gfx950) DynamicFusion hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its seam is the link-time set of
 * free functions kfusion::device::* declared in the private header
 * /root/reference/kfusion/src/internal.hpp:104-115 and called from the public class
 * kfusion::cuda::TsdfVolume (kfusion/src/tsdf_volume.cpp).  Every entry point below names the
 * reference interface it replaces.  Plain C types only: raw device pointers, byte pitches,
 * row-major float arrays; no torch / OpenCV / HIP types in any signature (a stream is an opaque
 * void* holding a hipStream_t; NULL = the default stream).
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a hipError_t value or a DF_E_* code;
 *     nothing prints, nothing calls exit() (the reference's cudaSafeCall prints and exit(0)s,
 *     kfusion/src/safe_call.hpp:13-27; the C++ wrapper maps non-zero to kfusion::cuda::error).
 *   - pointers named *_dev are DEVICE pointers; small parameter blocks (affines, intrinsics)
 *     are HOST pointers read before the call returns (the reference passes them by value).
 *   - an affine is 12 floats: R row-major [9] then t [3]  (device::Aff3f, internal.hpp:26-27,
 *     filled by device_cast, kfusion/src/precomp.hpp:19-28).
 *   - kernels are enqueued on `stream` and NOT synchronised (the reference's integrate ends in
 *     cudaDeviceSynchronize, tsdf_volume.cu:160; the C++ wrapper restores that behaviour).
 *   - no global state: any number of volumes and warp fields may be used concurrently, each from its own stream
 *     (the reference is not re-entrant: global texture ref tsdf_volume.cu:50, host globals
 *     warp_field.cpp:11-15).  ONE warp-field handle, though, is single-stream: its calls rewrite
 *     scratch it owns (the point queries' fallback list, the solver workspace, the cull's device
 *     scalars, the dists max-pyramid and the launch plan of the warped sweep), so calls on the same
 *     DfWarpField must be issued on one stream or serialised by the caller.  (The handle also owns one
 *     internal side stream, on which dfusion_integrate_warped makes look-ahead tables beside its sweep;
 *     the next call on the handle waits for it on the device -- invisible to the caller.)  dfusion_integrate keeps
 *     its pyramid and launch plan in a scratch buffer cached per (device, stream) -- calls on one
 *     stream are ordered, calls on different streams use different buffers; dfusion_release_scratch()
 *     frees them.  At most 8 buffers are kept per device (the least recently used idle one goes); a
 *     buffer is held for the whole of the call that uses it, so host threads may call on different
 *     streams concurrently (two threads on the SAME stream are serialised for the enqueue).
 *     Validation switches and measurement counters are per call / per handle (ABI 4).
 */
#ifndef DFUSION_H
#define DFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFUSION_ABI_VERSION 7   /* 7: dfusion_raycast_min_pieces (the sharded cast's key merge as direct exchanges: all-to-all of row bands, local MIN, all-gather); 6: dfusion_cloud_to_depth; the planned warped sweep takes every node count (no LDS node table: DF_WARP_NO_LDS now only selects the plain gather kernel), a prepared plan is also voided by set_nodes / build_index / a second set_transforms; 5: dfusion_integrate_warped_prepare / _sweep (the frame's integrate in two calls, for cross-frame overlap), dfusion_raycast_sum_pieces (direct row-band merge), DF_WARP_STEADY_PREFETCH, dfusion_selftest_exact_forms takes TEN counters ([8], [9]: the f32-division form of the blend's first normalisation), DF_WARP_NO_CODES + dfusion_warp_coded_blocks (4-bit neighbour codes of modelled blocks); 4: no process-wide state left: dfusion_integrate_ex (validation flags + swept counter per call) replaces dfusion_debug_rigid / dfusion_debug_rigid_counters, dfusion_warp_debug_counters (per handle) replaces dfusion_debug_warp_counters; dfusion_warp_alive_blocks; dfusion_raycast_points_of_keys_rows; DF_WARP_NO_PREFETCH; 3: dfusion_raycast_points_of_keys (dfusion_raycast_shade's points nullable), dfusion_release_scratch, DF_INDEX_TABLES_ON_DEMAND, DF_WARP_*_BLOCK_MODEL flags; 2: sharded cast merges on one key (no vertex exchange), dfusion_debug_rigid_counters, selftest counts[6] */

typedef void *dfStream; /* hipStream_t */

/* device::TsdfVolume, kfusion/src/internal.hpp:29-49, field for field.
 * Voxel = ushort2 {half tsdf, u16 weight} (device.hpp:53-61); linear index
 * x + y*dims[0] + z*dims[0]*dims[1] (device.hpp:17-18).                                      */
typedef struct DfVolume {
    void *data;          /* DEVICE pointer to the first STORED plane (see DfSlab)            */
    int dims[3];         /* global voxel counts (x, y, z); dims[0] % 4 == 0                  */
    float voxel_size[3]; /* metres                                                            */
    float trunc_dist;    /* metres                                                            */
    int max_weight;
} DfVolume;

/* Z-slab view for multi-GPU sharding (no reference counterpart: the reference is single-GPU).
 * `data` holds planes [z_store0, z_store0+z_store_n) (own planes + halos); this shard integrates
 * and owns ray steps for planes [z_own0, z_own0+z_own_n).  NULL = the whole volume.           */
typedef struct DfSlab {
    int z_store0, z_store_n;
    int z_own0, z_own_n;
} DfSlab;

/* Opaque warp-field handle: device copy of the deformation nodes (WarpField::nodes_,
 * kfusion/include/kfusion/warp_field.hpp:35-40) plus the k-NN brick index that replaces the
 * nanoflann kd-tree (warp_field.cpp:275-282).                                                 */
typedef struct DfWarpField DfWarpField;

enum {
    DF_OK = 0,
    DF_E_INVALID = 100001,   /* bad argument (null pointer, dims, k not in {1..8}, M < k, ...) */
    DF_E_NO_INDEX = 100002,  /* dfusion_warp_build_index not called for this geometry / k       */
    DF_E_NO_DEVICE = 100003  /* no HIP device / kernel image not loadable                       */
};

/* flags for dfusion_integrate_warped */
#define DF_WARP_NO_CULL 1u   /* disable the (result-identical) conservative brick culling     */
#define DF_WARP_NO_TABLE 2u  /* ignore the per-voxel k-NN table even if built (re-rank per frame) */
#define DF_WARP_NO_WEIGHT_TABLE 4u /* ignore the per-voxel weight table (recompute exp per frame)   */
#define DF_WARP_NO_LDS 8u    /* the plain gather kernel: node transforms from global memory per neighbour, no launch
                                plan, no verdict pass, no codes, no prepare / sweep split; validation switch.  (Until
                                round 6 this was also the path of node sets whose table exceeds the 160 KiB LDS,
                                M > 5120; the planned sweep now needs no LDS node table and takes every node count.) */
#define DF_WARP_NO_PIPELINE 16u /* batched instead of software-pipelined table loads; validation switch */
#define DF_WARP_NO_ZERO_SKIP 32u /* also sweep tiles whose blend weights are all so small that the reference's
                                 normalisation divides by zero (they cannot update); validation switch  */
#define DF_WARP_NO_DEPTH_PYRAMID 64u /* cull against the image-wide maximum of dists only, not against the maximum
                                 over the pixels a tile can project to; validation switch                  */
#define DF_WARP_NO_BLOCK_MODEL 128u /* launch plan from the ball test alone: do not build / use the per-block blend models
                                 (bounds of each 8x8x8 block's blend weights, made from the weight table for the blocks
                                 a sweep finds alive, from the second sweep over a table on; 10 bytes x 16 per block);
                                 validation switch                                                              */
#define DF_WARP_BLOCK_MODEL_NOW 256u /* make block models from the FIRST sweep over a new weight table on (default: the
                                 second, so that a node set that changes every frame never pays for them)       */
#define DF_WARP_NO_PREFETCH 512u /* everything on the caller's stream, in order: no look-ahead builds on the handle's side stream
                                 (by default the tables / blend models of blocks NEAR the frame's alive set -- what a moving camera
                                 or a changing warp brings in over the next few frames -- are made beside the sweep, on a stream the
                                 handle owns, so that a block is usually built before it is first swept); validation switch     */
#define DF_WARP_NO_CODES 2048u /* the sweep reads the 16-byte neighbour-index record of every voxel and gathers its neighbours' transforms
                                 * from global memory, instead of the 4-bit codes into the per-wave copies of the 4 x 4 x 4 sub-block unions
                                 * (k = 8; every block the model pass has visited); validation / A/B switch (ABI 5)                      */
#define DF_WARP_STEADY_PREFETCH 1024u /* keep the look-ahead side stream on in EVERY frame.  By default a handle whose last plan-kernel report
                                 * listed nothing to build switches it off until the next probe (every 8th sweep); that report is read from
                                 * pinned host memory WITHOUT a synchronisation, so which frame switches depends on host / GPU timing --
                                 * never any result, but swept-voxel counters and frame times of a run.  For tests and measurements that
                                 * compare such counters between runs (ABI 5).                                                         */
/* flags for dfusion_warp_build_index */
#define DF_INDEX_VOXEL_TABLE 1u /* also cache the exact k-NN of EVERY voxel of the slab in HBM:
                                   k * 2 bytes per voxel (2 GiB at 512^3, k = 8) -- the per-frame
                                   sweep then streams it instead of re-running the k-NN search   */
#define DF_INDEX_WEIGHT_TABLE 2u /* implies the above, and also caches the k blend weights
                                   exp(-d^2/(2 dg_w^2)) of every voxel (k * 4 bytes per voxel, 4 GiB
                                   at 512^3, k = 8): they too depend only on canonical geometry     */
#define DF_INDEX_TABLES_ON_DEMAND 4u /* with DF_INDEX_WEIGHT_TABLE: allocate the tables but fill them 8x8x8 block by
                                   block, when a warped integrate's launch plan first finds a block alive (a frame
                                   sweeps a third of the volume; the build is the cost of a node-set change).
                                   Sweeps that take no verdicts (DF_WARP_NO_CULL, DF_WARP_NO_PIPELINE, ...) build what
                                   is missing first.  Results are identical either way                          */

int dfusion_abi_version(void);
const char *dfusion_error_string(int err);

/* ---- volume -------------------------------------------------------------------------------
 * device::clear_volume (internal.hpp:105; tsdf_volume.cu:15-41): every stored voxel <- 0.     */
int dfusion_clear(DfVolume v, const DfSlab *slab, dfStream stream);

/* device::compute_dists (internal.hpp:124; kfusion/src/cuda/imgproc.cu:259-294): depth mm (u16)
 * -> ray length in metres as IEEE-half bits.  intr = {fx, fy, cx, cy}.                         */
int dfusion_compute_dists(const uint16_t *depth_dev, size_t depth_pitch, uint16_t *dists_dev, size_t dists_pitch,
                          int cols, int rows, const float intr[4], dfStream stream);

/* device::project_and_remove / device::project (internal.hpp:107-109; project_kernel tsdf_volume.cu:113-139, launched
 * by :163-192) fused with the per-point arithmetic of TsdfVolume::psdf (tsdf_volume.cpp:266-292).
 * points_dev: n float4 (camera frame), updated in place: NaN points untouched; points projecting outside the image ->
 * (qnan, qnan, qnan, 0); otherwise (coo.x*Dp, coo.y*Dp, Dp, 0) with Dp = dists(coo) and the dists pixel "removed" (<- 0).
 * The reference samples and zeroes ONE image concurrently (a second point on the same pixel may see 0 or the old
 * value); here samples come from dists_in and the zeros go to dists_out (nullable; must not alias dists_in; the caller
 * seeds it with a copy of dists_in).  ro_dev (nullable): n floats, psdf's return value
 * (K^-1 * new_point)[2] - old_point.z, NaN for NaN / outside points.  n_inside_dev (nullable) is INCREMENTED by the
 * number of points that landed inside the image.                                                                  */
int dfusion_project_and_remove(const uint16_t *dists_in_dev, size_t in_pitch, uint16_t *dists_out_dev, size_t out_pitch,
                               int cols, int rows, float *points_dev, unsigned long long n, const float proj[4],
                               float *ro_dev, unsigned long long *n_inside_dev, dfStream stream);

/* device::integrate (internal.hpp:106; tsdf_volume.cu:51-112,141-161): rigid projective TSDF
 * update.  proj = {fx, fy, cx, cy} (device::Projector).  n_updated_dev (nullable) is
 * INCREMENTED by the number of voxels whose update branch (tsdf_volume.cu:91) was taken.
 * Scratch: a launch plan, the chunk starts of every column patch (16 bytes per column and 32-plane chunk: 64 MiB at 512^3, 512 MiB at
 * 1024^3), the plan's bins and a max-pyramid of `dists`, + 25 % growth headroom -- about 84 MB at 512^3, 0.65 GB at 1024^3 -- in a
 * buffer the library keeps per (device, stream) and grows on demand.  It is KEPT until dfusion_release_scratch(); at most 8
 * (device, stream) pairs are cached, the least recently used one is evicted (the runtime's stream-ordered allocator was tried and
 * gave wrong plans in processes that also hipMalloc / hipFree between the calls).  Limit: (columns / 64, rounded up to whole patches) x
 * (32-plane chunks of the slab) < 2^30, i.e. any volume that fits the device.                                                   */
int dfusion_integrate(const uint16_t *dists_dev, size_t dists_pitch, int cols, int rows, DfVolume v,
                      const DfSlab *slab, const float vol2cam[12], const float proj[4],
                      unsigned long long *n_updated_dev, dfStream stream);

/* The same with the validation switches and the measurement counter of THIS call (no reference counterpart; tests assert the
 * volumes are identical with and without each switch).  flags = 0: everything on.  n_swept_dev (nullable, device, 8 bytes) is
 * INCREMENTED by the number of voxels the sweep put through the projective sample (tsdf_volume.cu:77-93) -- the voxels of the
 * launch plan's alive sub-chunks; beside n_updated_dev this gives swept / updated, the sweep's over-work.                        */
#define DF_RIGID_NO_DEPTH_CULL 1u  /* no behind-the-surface test (a conservative skip of voxels more than trunc_dist behind every
                                      depth value they can be compared with; per-frame max-pyramid of dists)                     */
#define DF_RIGID_NO_SHORT_FORMS 2u /* the generic correctly rounded divisions / square root everywhere                           */
#define DF_RIGID_KEEP_ALL 4u       /* the launch plan keeps every sub-chunk (no frustum test either): every voxel goes through the
                                      reference's own tests                                                                       */
#define DF_RIGID_NO_SAT 8u         /* no saturated-sample shortcuts (batches of voxels all farther than trunc_dist from the surface
                                      skip the exact square root and, onto stored 1.0 / cleared voxels, the fuse division)        */
#define DF_RIGID_POISON_SCRATCH 16u /* fill the call's scratch (launch plan, chunk starts, pyramid) with 0xFF bytes first: the buffer is
                                      kept between calls, and a read of plan data THIS call did not write would otherwise see the
                                      previous call's valid-looking values                                                        */
int dfusion_integrate_ex(const uint16_t *dists_dev, size_t dists_pitch, int cols, int rows, DfVolume v,
                         const DfSlab *slab, const float vol2cam[12], const float proj[4], unsigned flags,
                         unsigned long long *n_updated_dev, unsigned long long *n_swept_dev, dfStream stream);

/* Frees the scratch buffers dfusion_integrate keeps per (device, stream), after a device synchronise (a cached stream handle may no
 * longer exist); the caller's current device is restored.  Optional; for hosts that tear devices down or count allocations.       */
int dfusion_release_scratch(void);

/* device::raycast, Points variant (internal.hpp:113-114; tsdf_volume.cu:340-405,459-474).
 * points/normals: float4 per pixel, byte pitches, misses = all-NaN.  reproj = {1/fx, 1/fy, cx,
 * cy} (device::Reprojector, precomp.cpp:55).  keys_dev (nullable, cols*rows uint32): per-pixel
 * first-event key (step<<1 | hit) on steps this slab owns, 0xffffffff = none (sharded merge).  */
int dfusion_raycast_points(DfVolume v, const DfSlab *slab, const float cam2vol[12], const float Rinv[9],
                           const float reproj[4], float *points_dev, size_t points_pitch, float *normals_dev,
                           size_t normals_pitch, int cols, int rows, float step_factor, float delta_factor,
                           uint32_t *keys_dev, dfStream stream);

/* device::raycast, Depth variant (internal.hpp:110-111; tsdf_volume.cu:272-338,441-456).       */
int dfusion_raycast_depth(DfVolume v, const DfSlab *slab, const float cam2vol[12], const float Rinv[9],
                          const float reproj[4], uint16_t *depth_dev, size_t depth_pitch, float *normals_dev,
                          size_t normals_pitch, int cols, int rows, float step_factor, float delta_factor,
                          dfStream stream);

/* Z-slab (multi-GPU) cast in two stages; no reference counterpart.  The zero-crossing refinement
 * Ts = t - step*Ft/(Ftdt-Ft) (tsdf_volume.cu:389) may EXTRAPOLATE arbitrarily far along the ray, so the
 * vertex of a hit can lie in another GPU's slab: stage 1 finds, on the steps this slab owns, the first
 * event and (for hits) Ts; the host MIN-merges the 64-bit keys over the ranks -- ONE collective, which
 * delivers the first event along every ray, its owner and its Ts; stage 2 lets the slab that owns the
 * vertex' nearest plane compute the normal (needs a 2-plane halo) and write the final camera-frame
 * point/normal, with vertex = origin + direction * Ts recomputed from the pixel exactly as the unsharded
 * cast computes it (:390).  Pixels a slab does not resolve are written as all-zero BITS (the slab owning
 * plane 0 writes the NaN fill of misses), so integer-summing the slabs' outputs equals the unsharded cast.
 *
 * key layout (a non-negative int64, so ncclMin on ncclInt64 orders it):
 *   bit 63 = 0 | bits 62..40 step index k | bit 39 kind (1 = hit, 0 = back-face break) | bits 38..32 rank_tag | bits 31..0 Ts (f32 bits)
 * DF_RC_KEY_NONE (no event on this slab's steps) is larger than every event key.  dfusion_raycast_march returns DF_E_INVALID when
 * (volume diagonal) / (trunc_dist * step_factor) could reach 2^23 steps -- the index would not fit its field.                   */
#define DF_RC_KEY_NONE 0x7fffffffffffffffull
#define DF_RC_KEY_MAX_RANK 127u
int dfusion_raycast_march(DfVolume v, const DfSlab *slab, const float cam2vol[12], const float reproj[4], int cols,
                          int rows, float step_factor, unsigned int rank_tag, unsigned long long *keys64_dev,
                          dfStream stream);
int dfusion_raycast_shade(DfVolume v, const DfSlab *slab, const float cam2vol[12], const float Rinv[9],
                          const float reproj[4], const unsigned long long *merged_keys64_dev,
                          float *points_dev, size_t points_pitch, float *normals_dev, size_t normals_pitch, int cols,
                          int rows, float delta_factor, dfStream stream);
/* Stage 3, on the rank that wants the image: the camera-frame POINTS from the merged keys and the summed normals -- they need no
 * exchange (the vertex is origin + direction * Ts, and a hit stands iff the owner's normal is not the NaN fill: 4th component 0), so
 * dfusion_raycast_shade may be given points_dev = NULL and only the normals cross GPUs.  Equals the points image of the unsharded
 * cast bit for bit.  Needs no volume.                                                                                            */
int dfusion_raycast_points_of_keys(const float cam2vol[12], const float Rinv[9], const float reproj[4],
                                   const unsigned long long *merged_keys64_dev, const float *normals_dev, size_t normals_pitch,
                                   float *points_dev, size_t points_pitch, int cols, int rows, dfStream stream);
/* The same for a BAND of pixel rows [row0, row0 + rows) of a cols x image_rows image: merged_keys64_dev is the whole image's (every
 * rank holds it after the merge), normals_dev / points_dev point at the band's first row.  For the row-banded merge: the normals are
 * reduce-scattered by pixel rows, every rank finishes its own band, and no rank receives the whole image (ABI 4).              */
int dfusion_raycast_points_of_keys_rows(const float cam2vol[12], const float Rinv[9], const float reproj[4],
                                        const unsigned long long *merged_keys64_dev, const float *normals_dev, size_t normals_pitch,
                                        float *points_dev, size_t points_pitch, int cols, int image_rows, int row0, int rows,
                                        dfStream stream);

/* The receiving end of the DIRECT form of the sharded cast's second collective (ABI 5): rank r gets, from every rank p, p's piece of r's row
 * band of the normals image (fixed-size pieces: one all-to-all, no counts; every pixel's normal is non-zero in exactly one piece) and adds
 * them: out[i] = sum over p < n_pieces of pieces[p * n_words + i], 32-bit integer adds on the bit patterns (all summands but one are zero, so
 * the sum IS the owner's bits -- the same arithmetic a reduce_scatter(SUM) on the int32 view performs).  n_words: 32-bit words per piece.   */
int dfusion_raycast_sum_pieces(const uint32_t *pieces_dev, int n_pieces, unsigned long long n_words, uint32_t *out_dev, dfStream stream);

/* The local step of the DIRECT form of the sharded cast's FIRST collective (ABI 7).  ncclAllReduce(MIN) of the key image walks a ring of
 * 2 (N - 1) sequential steps; xGMI links every pair of GPUs of a node directly, so the same result takes two exchanges of ONE step each:
 * every rank sends rank r its copy of r's band of pixel rows (one all-to-all of fixed-size pieces, as for the normals), r takes the per-key
 * minimum of the N pieces -- this call -- and one all-gather hands every rank the merged image.  out[i] = min over p < n_pieces of
 * pieces[p * n_keys + i]; keys are the non-negative int64 merge keys above (DF_RC_KEY_NONE in padding rows).  n_keys even unless
 * n_pieces == 1; pieces_dev / out_dev 16-byte aligned.                                                                                */
int dfusion_raycast_min_pieces(const unsigned long long *pieces_dev, int n_pieces, unsigned long long n_keys, unsigned long long *out_dev, dfStream stream);

/* ---- surface extraction -------------------------------------------------------
 * device::extractCloud (internal.hpp:142; tsdf_volume.cu:511-710,798-817): zero crossings between every voxel and its
 * +x/+y/+z neighbour, linearly interpolated, transformed by aff (= volume pose).  points_dev: float4 x capacity;
 * *count_dev (device, zero it first) is INCREMENTED by the number of crossings found -- the number written is
 * min(count, capacity); output order is unspecified (as in the reference).  A slab needs one halo plane above.  */
int dfusion_extract_cloud(DfVolume v, const DfSlab *slab, const float aff[12], float *points_dev,
                          unsigned long long capacity, unsigned long long *count_dev, dfStream stream);
/* device::extractNormals (internal.hpp:143; tsdf_volume.cu:714-795,819-831): TSDF gradient at each extracted point. */
int dfusion_extract_normals(DfVolume v, const DfSlab *slab, const float aff[12], const float Rinv[9],
                            const float *points_dev, unsigned long long n, float gradient_delta_factor,
                            float *normals_dev, dfStream stream);

/* ---- warp field -----------------------------------------------------------------------------
 * WarpField::WarpField / ~WarpField (warp_field.cpp:17-34).                                    */
int dfusion_warp_create(DfWarpField **out);
int dfusion_warp_destroy(DfWarpField *wf);

/* WarpField::init + buildKDTree (warp_field.cpp:41-88,275-282): upload M nodes.
 *   pos_dev[M*3]   deformation_node::vertex
 *   dq_dev[M*8]    deformation_node::transform = {rotation_ (w,x,y,z), translation_ (w,x,y,z)},
 *                  the first 32 bytes of utils::DualQuaternion<float> (dual_quaternion.hpp:231-232)
 *   sigma_dev[M]   deformation_node::weight (dg_w)
 * Invalidates the k-NN index (node positions changed).                                        */
int dfusion_warp_set_nodes(DfWarpField *wf, const float *pos_dev, const float *dq_dev, const float *sigma_dev,
                           int M, dfStream stream);

/* Per-frame transform update (what WarpFieldOptimiser writes back, CombinedSolver.h:189-197);
 * node positions, hence the k-NN index, are unchanged.                                        */
int dfusion_warp_set_transforms(DfWarpField *wf, const float *dq_dev, dfStream stream);

/* Builds the exact k-NN acceleration index for voxel queries of this volume geometry: for each
 * 8x8x8 brick the list of nodes that can be among the k nearest of ANY of its voxels, and (flag
 * DF_INDEX_VOXEL_TABLE) the per-voxel k-NN table of the slab's own planes.  Replaces the kd-tree
 * build (warp_field.cpp:275-282); needed again only when node POSITIONS change (they do not between
 * frames: only transforms are optimised).  `geometry.data` is not dereferenced.  Blocks until built. */
int dfusion_warp_build_index(DfWarpField *wf, DfVolume geometry, const DfSlab *slab, const float vol2world[12], int k,
                             unsigned flags, dfStream stream);

/* Introspection of the k-NN index (blocking): total candidate-list entries, number of 8^3 bricks, k it was built for. */
int dfusion_warp_index_info(const DfWarpField *wf, unsigned long long *total_entries, unsigned int *n_bricks, int *k_built);

/* Locality hint for dfusion_knn / dfusion_warp_points (and the solver's k-NN): the query points are the pixels of an image
 * `image_cols` wide in row-major order (the ray-cast cloud KinFu::dynamicfusion warps, kinfu.cpp:353-391).  Queries whose count is a
 * whole number of 8-row bands are then processed in 8 x 8 pixel tiles per wave instead of 64-pixel row segments; results are
 * identical, only faster (fewer distinct index bricks per wave).  0 switches it off (default).                                   */
int dfusion_warp_set_point_tiling(DfWarpField *wf, int image_cols);

/* Device self-test of the sweeps' short arithmetic forms (dfusion_device.h) against the generic ones they replace; no reference
 * counterpart, used by the parity tests.  counts_dev: TEN device entries (ABI 5; eight in ABI 2-4, six in ABI 1), cleared by the call, receive
 * mismatch counts: [0] short sqrtf over every f32 of its domain, [1] short f64 reciprocal over every positive normal f32, [2] packed
 * quaternion products on n_random random pairs (specials included), [3] near-unit normalisation on the normalised quaternions among
 * them, [4] how many of those there were, [5] the short fuse division on every finite stored half x 97 weights x (n_random >> 21)
 * tsdf values, [6] the projective sample -- tsdf_sample_fast and the two-stage saturation form of the rigid sweep -- against the
 * generic statements on n_random positions inside the forms' domain and on its edges, [7] how many of those samples updated,
 * [8] the blend's first normalisation as an f32 division (q_div_f32) against (float)((1.0 / (double)n) * (double)c) on the quaternions
 * its domain test accepts out of n_random random and edge-of-domain ones, [9] how many quaternions that were.                       */
int dfusion_selftest_exact_forms(unsigned long long n_random, unsigned long long *counts_dev, dfStream stream);

/* Measurement hook of dfusion_integrate_warped's cached sweep, per warp-field handle (NULL switches it off): while set, every
 * launch through this handle ADDS to *swept_dev (device, 8 bytes) the voxels of the (8 x 8 column patch, 8-plane layer) cells its
 * launch plan keeps, i.e. the voxels that go through blend -> transform -> project.                                              */
int dfusion_warp_debug_counters(DfWarpField *wf, unsigned long long *swept_dev);

/* What the last dfusion_integrate_warped through this handle found alive: per_layer_dev[l] (device, n_layers entries, l = global
 * plane / 8) is INCREMENTED by the number of 8 x 8 x 8 blocks of layer l that its verdict pass kept, for the layers that lie
 * entirely inside planes [z0, z0 + zn).  Z-slab re-balancing reads it (one all-reduce over the ranks gives the global profile of the
 * sweep's real work per plane).  DF_E_NO_INDEX when no sweep with verdicts has run since the index was built.                       */
int dfusion_warp_alive_blocks(DfWarpField *wf, int z0, int zn, unsigned long long *per_layer_dev, int n_layers, dfStream stream);
/* Same, counting only the kept blocks that had 4-bit neighbour codes for the sweep to read (blocks with a blend model made in an
 * earlier frame; DF_WARP_NO_CODES makes a sweep ignore them, the count is of what was available).  A measurement hook: tests use it
 * to see that the coded path engaged.                                                                                              */
int dfusion_warp_coded_blocks(DfWarpField *wf, int z0, int zn, unsigned long long *per_layer_dev, int n_layers, dfStream stream);

/* WarpField::KNN (warp_field.cpp:247-251) for N query points [N*3]: idx[N*k] int32, d2[N*k], ascending distance; exactly
 * equidistant nodes in the order the reference's nanoflann walk meets them (nanoflann.hpp:110-131,1200-1254).                    */
int dfusion_knn(DfWarpField *wf, int k, const float *queries_dev, int N, int *idx_dev, float *d2_dev, dfStream stream);

/* WarpField::warp (warp_field.cpp:180-195) on device: points/normals [N*3] in place
 * (normals_dev nullable); NaN points are skipped; warp_to_live = WarpField::warp_to_live_.     */
int dfusion_warp_points(DfWarpField *wf, int k, float *points_dev, float *normals_dev, int N,
                        const float warp_to_live[12], dfStream stream);

/* WarpFieldOptimiser::optimiseWarpData (CombinedSolver / Opt energy kfusion/solvers/dynamicfusion.t:26-52) and
 * WarpField::energy_data (Ceres, warp_field.cpp:117-163, functor optimisation.hpp:36-71): the DATA term
 *     E(T) = sum_v | (live_v - canonical_v) - sum_{i<k} w_vi * T_{n_vi} |^2
 * over the node translations T (n_vi, w_vi: the k nearest nodes of canonical_v and their weights; rotations are not
 * unknowns and there is no regularisation term in the reference either).  Linear least squares: `iters` conjugate-
 * gradient steps on (W^T W + lambda I) delta = W^T e(T_now), all on the device, then T <- T_now + delta and every node's
 * translation_ <- 0.5 * (0, T) * rotation_ (encodeTranslation, dual_quaternion.hpp:82-85) -- the node transforms of `wf`
 * are updated in place (the k-NN index stays valid: positions are untouched).
 *   canonical_dev, live_dev  [N*3] packed; a NaN component in either skips the point (warp_field.cpp:130-136)
 *   dq_out_dev (nullable)    [M*8] the updated transforms, layout of dfusion_warp_set_nodes
 *   energy_dev (nullable)    2 floats: E before, E after
 * Float sums use fixed trees (no atomics): the result is reproducible run to run.                                      */
int dfusion_warp_solve_data_term(DfWarpField *wf, int k, const float *canonical_dev, const float *live_dev, int N, int iters,
                                 float lambda, float *dq_out_dev, float *energy_dev, dfStream stream);

/* The north-star kernel: per-voxel DQB (WarpField::DQB, warp_field.cpp:203-217) composed with
 * TsdfIntegrator (tsdf_volume.cu:77-104): x_c = vol2world*voxel, x_w = DQB(x_c).transform(x_c),
 * vc = world2cam*x_w, then the projective update.  Requires dfusion_warp_build_index.         */
int dfusion_integrate_warped(const uint16_t *dists_dev, size_t dists_pitch, int cols, int rows, DfVolume v,
                             const DfSlab *slab, const float vol2world[12], const float world2cam[12],
                             const float proj[4], DfWarpField *wf, int k, unsigned flags,
                             unsigned long long *n_updated_dev, dfStream stream);

/* The same frame in TWO calls (ABI 5), for hosts that pipeline frames: everything of the warped integrate that does not touch the volume --
 * the dists max-pyramid, the per-block verdict pass, on-demand table / model builds, the launch plan: ~70 us of small kernels at the
 * headline size -- and the sweep.  `prepare` (and the dfusion_warp_set_transforms before it) may be issued on another stream than `sweep` and
 * then runs BESIDE the previous frame's sweep and ray-cast: what a sweep reads of the handle -- the node transform arrays, the launch plan --
 * is double-buffered, and the handle orders the rest itself (a sweep waits for its prepare on the device; set_transforms / prepare wait
 * for the sweep TWO frames back, whose buffers they reuse).  The caller owns the dists image: give consecutive frames different buffers.
 * All sweeps of a handle go on ONE stream; every other call on the handle must still be ordered by the caller, and a handle is driven
 * either through these two calls or through dfusion_integrate_warped on one stream (mixing them is safe only on that one stream).
 * Results are those of dfusion_integrate_warped, bit for bit.  Only the cached path has a plan to
 * prepare: DF_E_INVALID without the per-voxel weight tables, with DF_WARP_NO_PIPELINE / NO_LDS / NO_TABLE, or k other than 4 / 8.
 * geometry.data is not dereferenced by prepare (may be NULL); sweep wants the same dims / voxel size / slab.  A prepared plan is void
 * -- its sweep call returns DF_E_INVALID -- after any other integrate on the handle, after dfusion_warp_set_nodes or
 * dfusion_warp_build_index (they free or re-make what the plan points at), and after a SECOND dfusion_warp_set_transforms since the
 * prepare (the first writes the alternate node set; the second would rewrite the one the plan reads).                                 */
int dfusion_integrate_warped_prepare(const uint16_t *dists_dev, size_t dists_pitch, int cols, int rows, DfVolume geometry,
                                     const DfSlab *slab, const float vol2world[12], const float world2cam[12], const float proj[4],
                                     DfWarpField *wf, int k, unsigned flags, dfStream stream);
int dfusion_integrate_warped_sweep(DfVolume v, const DfSlab *slab, DfWarpField *wf, unsigned long long *n_updated_dev, dfStream stream);


/* ---- depth front-end + projective ICP -----------------------------------------------------
 * All images are pitched device memory: depth u16 mm, points / normals float4.  `intr` = {fx, fy, cx, cy} of the
 * pyramid LEVEL the images belong to (Intr::operator()(level) / setLevelIntr divide by 2^level).
 * The reference's __expf (bilateral weight) and rsqrt (normalisation) are hardware approximations; this library uses
 * (float)exp((double)x) and 1/sqrtf -- see oracle/dfusion_frontend_oracle.c.                                          */

/* device::bilateralFilter (internal.hpp:128; kfusion/src/cuda/imgproc.cu:11-59).  sigma_depth in metres. src != dst. */
int dfusion_bilateral_filter(const uint16_t *src_dev, size_t src_pitch, uint16_t *dst_dev, size_t dst_pitch, int cols, int rows,
                             int kernel_size, float sigma_spatial, float sigma_depth, dfStream stream);
/* device::truncateDepth (internal.hpp:127; imgproc.cu:66-85): depth > max_dist (metres) <- 0, in place.            */
int dfusion_truncate_depth(uint16_t *depth_dev, size_t pitch, int cols, int rows, float max_dist, dfStream stream);
/* device::cloud_to_depth (internal.hpp:125; imgproc.cu:273-282, 296-303), behind cuda::cloudToDepth (imgproc.cpp:98-103): depth (mm) =
 * points.z (metres) * 1000, float -> ushort toward zero and saturating, NaN (a ray-cast miss) -> 0 (ABI 6).                          */
int dfusion_cloud_to_depth(const float *points_dev, size_t points_pitch, uint16_t *depth_dev, size_t depth_pitch, int cols, int rows, dfStream stream);
/* device::depthPyr (internal.hpp:129; imgproc.cu:94-137): dst is (src_rows/2) x (src_cols/2).                       */
int dfusion_depth_pyramid(const uint16_t *src_dev, size_t src_pitch, int src_cols, int src_rows, uint16_t *dst_dev, size_t dst_pitch,
                          float sigma_depth, dfStream stream);
/* device::computeNormalsAndMaskDepth (internal.hpp:134; imgproc.cu:145-201): normals (x,y,z,0) or (qnan,qnan,qnan,0);
 * depth pixels without a normal are zeroed in place.                                                                 */
int dfusion_compute_normals_mask_depth(uint16_t *depth_dev, size_t depth_pitch, float *normals_dev, size_t normals_pitch, int cols,
                                       int rows, const float intr[4], dfStream stream);
/* device::computePointNormals (internal.hpp:135; imgproc.cu:210-252): invalid pixels are all-NaN in both outputs.    */
int dfusion_compute_point_normals(const uint16_t *depth_dev, size_t depth_pitch, float *points_dev, size_t points_pitch,
                                  float *normals_dev, size_t normals_pitch, int cols, int rows, const float intr[4], dfStream stream);
/* device::resizeDepthNormals / resizePointsNormals (internal.hpp:131-132; imgproc.cu:309-414): outputs are half size. */
int dfusion_resize_depth_normals(const uint16_t *depth_dev, size_t depth_pitch, const float *normals_dev, size_t normals_pitch,
                                 int src_cols, int src_rows, uint16_t *depth_out_dev, size_t depth_out_pitch, float *normals_out_dev,
                                 size_t normals_out_pitch, dfStream stream);
int dfusion_resize_points_normals(const float *points_dev, size_t points_pitch, const float *normals_dev, size_t normals_pitch,
                                  int src_cols, int src_rows, float *points_out_dev, size_t points_out_pitch, float *normals_out_dev,
                                  size_t normals_out_pitch, dfStream stream);

/* device::renderImage (Points and Depth variants) and renderTangentColors (internal.hpp:134-136; kernels imgproc.cu:420-583): the
 * Phong view KinFu::renderImage produces (kinfu.cpp:312-343,408-436).  image: cols x rows BGRA bytes (device), light_pose in metres
 * (KinFuParams::light_pose), intr = (fx, fy, cx, cy) of the depth image (the Points variant shades the points as given).             */
int dfusion_render_image_points(const float *points_dev, size_t points_pitch, const float *normals_dev, size_t normals_pitch, int cols,
                                int rows, const float light_pose[3], unsigned char *image_dev, size_t image_pitch, dfStream stream);
int dfusion_render_image_depth(const uint16_t *depth_dev, size_t depth_pitch, const float *normals_dev, size_t normals_pitch, int cols,
                               int rows, const float intr[4], const float light_pose[3], unsigned char *image_dev, size_t image_pitch,
                               dfStream stream);
int dfusion_render_tangent_colors(const float *normals_dev, size_t normals_pitch, int cols, int rows, unsigned char *image_dev,
                                  size_t image_pitch, dfStream stream);

/* The host loops of KinFu::dynamicfusion (kinfu.cpp:353-383) on the device: out(y,x) = aff * in(y,x) for a rows x cols grid of
 * 3-vectors (aff nullable = plain re-striding), with cv::Affine3f * Vec3f float arithmetic (products summed left to right,
 * then + t).  Element strides in floats (input >= 3, output 3 or 4; a 4th output component is set to 0), row pitches in
 * bytes; a flat list of n points is rows = 1, cols = n.  NaN components propagate.  in != out.                        */
int dfusion_transform_points(const float *in_dev, size_t in_pitch, int in_stride, float *out_dev, size_t out_pitch, int out_stride,
                             int cols, int rows, const float aff[12], dfStream stream);

/* device::ComputeIcpHelper::operator() (internal.hpp:91-92; kfusion/src/cuda/proj_icp.cu:30-441): one Gauss-Newton
 * accumulation of point-to-plane ICP.  aff = current estimate curr -> prev; dist2_thres = dist_thres^2 and
 * min_cosine = cos(angle_thres) (projective_icp.cpp:11-15).  sums_dev[27] = the upper triangle of A (6x6) interleaved
 * with b exactly as StreamHelper::get unpacks it (projective_icp.cpp:43-61): for i in 0..5, for j in i..6.
 * The float sums are taken over the reference's reduction tree (32x8-pixel blocks, strides 128..1, then 256 strided
 * partial sums), so they are reproducible bit for bit.  workspace_dev: dfusion_icp_workspace_floats(cols, rows) floats
 * (ComputeIcpHelper::allocate_buffer, proj_icp.cu:446-464).  accepted_dev (nullable) is INCREMENTED by the number of
 * accepted correspondences.  The 6x6 solve stays on the host (cv::solve in the reference).                           */
int dfusion_icp_workspace_floats(int cols, int rows);
int dfusion_icp_sums_points(const float *vcurr_dev, size_t vcurr_pitch, const float *ncurr_dev, size_t ncurr_pitch,
                            const float *vprev_dev, size_t vprev_pitch, const float *nprev_dev, size_t nprev_pitch, int cols, int rows,
                            const float aff[12], const float intr[4], float dist2_thres, float min_cosine, float *workspace_dev,
                            float *sums_dev, int *accepted_dev, dfStream stream);
int dfusion_icp_sums_depth(const uint16_t *dcurr_dev, size_t dcurr_pitch, const float *ncurr_dev, size_t ncurr_pitch,
                           const uint16_t *dprev_dev, size_t dprev_pitch, const float *nprev_dev, size_t nprev_pitch, int cols, int rows,
                           const float aff[12], const float intr[4], float dist2_thres, float min_cosine, float *workspace_dev,
                           float *sums_dev, int *accepted_dev, dfStream stream);

/* ProjectiveICP::estimateTransform (projective_icp.cpp:129-213) as ONE enqueue: for every pyramid level from the coarsest,
 * `iters` times { correspondences + 27 sums (as dfusion_icp_sums_*), 6x6 solve, Tinc * estimate } with the estimate kept in
 * device memory -- no host round trip per iteration (the reference synchronises a stream and solves on the host each time).
 *   levels[l]     images of pyramid level l (curr / prev are float4 points, or u16 depth when depth_variant != 0)
 *   intr          level-0 intrinsics {fx, fy, cx, cy}; level l uses intr / 2^l (setLevelIntr)
 *   workspace_dev dfusion_icp_workspace_floats(level-0 cols, rows) + 27 floats
 *   state_dev     13 floats out: curr -> prev estimate {R[9] row-major, t[3]} and ok (1, or 0 once a normal matrix was singular:
 *                 |det| < 1e-15 or NaN -- estimateTransform returns false there)
 * The 6x6 solve is LU with partial pivoting in double (cv::solve DECOMP_SVD in the reference; OpenCV is not in its tree).    */
typedef struct DfIcpLevel {
    const void *curr; size_t curr_pitch; const float *ncurr; size_t ncurr_pitch;
    const void *prev; size_t prev_pitch; const float *nprev; size_t nprev_pitch;
    int cols, rows, iters;
} DfIcpLevel;
int dfusion_icp_estimate(const DfIcpLevel *levels, int n_levels, int depth_variant, const float intr[4], float dist2_thres,
                         float min_cosine, float *workspace_dev, float *state_dev, dfStream stream);

/* ---- measurement helper: plain device copy used as the MEASURED HBM roofline denominator ---- */
int dfusion_copy_bandwidth_probe(void *dst_dev, const void *src_dev, size_t bytes, dfStream stream);
/* read-only stream of `bytes` (sink4_dev: 4 writable device bytes): the measured denominator for scan kernels */
int dfusion_read_bandwidth_probe(const void *src_dev, size_t bytes, void *sink4_dev, dfStream stream);

#ifdef __cplusplus
}
#endif
#endif /* DFUSION_H */

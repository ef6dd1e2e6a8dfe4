// dfusion_internal.h -- host-side internals shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "dfusion.h"
#include "dfusion_device.h"
#include "dfusion_nanoflann.h"

#define DF_BRICK 8                    // k-NN index brick edge (voxels)
// Look-ahead margin of the warped sweep's verdict pass (metres): blocks that would be alive with every cull radius widened by this much
// get their tables / blend models made on the handle's side stream before a sweep needs them.  A camera turning 0.25 degrees per frame
// moves a voxel 2 m from its axis by 9 mm, the benchmark's warp amplitude moves the cull radii by up to 2 cm per frame: 5 cm is two to
// five frames of lead, and widens the set that is ever built by a few per cent.
#define DF_WARP_PREFETCH_MARGIN_M 0.05f
#define DF_WARP_PREFETCH_CAP 1024u    // look-ahead table builds per frame at most: about one round of the resident build grid (50-80 us beside the sweep)

#define DF_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return (int)e__; } while (0)
#define DF_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

static inline DfSlab df_slab_or_full(const DfVolume& v, const DfSlab* s)
{
    if (s) return *s;
    DfSlab f; f.z_store0 = 0; f.z_store_n = v.dims[2]; f.z_own0 = 0; f.z_own_n = v.dims[2]; return f;
}
static inline bool df_slab_valid(const DfVolume& v, const DfSlab& s)
{
    return s.z_store_n > 0 && s.z_own_n >= 0 && s.z_store0 >= 0 && s.z_store0 + s.z_store_n <= v.dims[2] &&
           s.z_own0 >= s.z_store0 && s.z_own0 + s.z_own_n <= s.z_store0 + s.z_store_n;
}
static inline bool df_volume_valid(const DfVolume& v)
{
    return v.data && v.dims[0] > 0 && v.dims[1] > 0 && v.dims[2] > 0 && (v.dims[0] % 4) == 0 &&
           v.voxel_size[0] > 0 && v.voxel_size[1] > 0 && v.voxel_size[2] > 0 && v.trunc_dist > 0;
}
static inline DfAff df_aff(const float a[12]) { DfAff r; memcpy(r.R, a, 36); memcpy(r.t, a + 9, 12); return r; }

// Device-side view of the warp field + brick index (passed to kernels by value).
struct DfWarpView {
    const float4* pos_sigma;   // [M] xyz = vertex, w = dg_w
    const float4* rot;         // [M] rotation_ (w,x,y,z)
    const float4* dual;        // [M] translation_ (w,x,y,z)
    const float4* node_t;      // [M] getTranslation() of the node (w,x,y,z), dual_quaternion.hpp:120-125
    const float4* rt;          // [2M] rot_j, node_t_j interleaved (32 bytes a node: what the sweep's union refills and uncoded cells gather)
    int M;
    // brick index
    const uint32_t* brick_off; // [nb+1]
    const uint16_t* brick_list;
    const float* brick_thr;    // [nb] list radius: nodes outside brick b's list are farther than this from its centre
    int bx, by, bz;            // brick grid over the GLOBAL volume
    DfNfView nf;               // nanoflann's tree over the nodes: orders exactly equidistant nodes (dfusion_nanoflann.h)
};

struct DfWarpField {
    int device;
    int M, cap;
    float4 *pos_sigma, *rot, *dual, *node_t;
    // index
    uint32_t* brick_off; uint32_t* brick_cnt; uint16_t* brick_list; float* brick_thr;
    size_t off_cap, list_cap;
    uint32_t* scan_tmp; size_t scan_cap;       // tile sums / tile offsets of the brick-count scan
    int bx, by, bz, k_built;
    int geom_dims[3]; float geom_vs[3]; float geom_aff[12];
    float geom_inv[12]; bool geom_inv_ok;      // world -> volume (locates the brick of a query point)
    bool index_valid;
    // per-voxel k-NN table (optional, DF_INDEX_VOXEL_TABLE)
    uint16_t* knn_tab; size_t knn_tab_cap;      // elements (uint16)
    int tab_z0, tab_zn, tab_k; bool tab_valid;
    float* w_tab; size_t w_tab_cap; bool w_tab_valid;   // per-voxel blend weights (optional, DF_INDEX_WEIGHT_TABLE)
    float* tile_wmax; size_t tile_wmax_cap;             // per table tile: max over its voxels of the weight sum (with w_tab)
    // device scalars for the conservative brick cull: [0] max |t_i|, [1] max sin(theta_i/2), [2] max dists
    float* bounds_dev;
    int max_phase;                    // which of bounds_dev[6], [7] this frame's capped pyramid leaves the image-wide maximum in
    // dfusion_integrate_warped_prepare / _sweep (round 5): the launch state a prepare call leaves for the sweep call (opaque here: the
    // argument structs live in dfusion_warp.hip), and the two events that order the halves when they are issued on different streams
    void* prep; bool prep_valid;
    hipEvent_t ev_prep_done, ev_sweep_done[2]; bool split_events;
    // What a sweep issued through the split API may still be READING when the next frame's set_transforms / prepare arrive on another
    // stream is double-buffered, so that they can run BESIDE that sweep instead of after it: the node transform arrays (rot / dual / node_t
    // and their alternates), the launch plan (two sets of mask + bins; FOUR counter sets, zeroed two frames ahead).  Sweeps are numbered;
    // a buffer remembers the last sweep that reads it, the ring of two events holds the last two sweeps (all sweeps on one stream).
    float4 *rot_alt, *dual_alt, *node_t_alt;
    float4 *rt, *rt_alt;                             // {rot, node_t} interleaved, and its alternate (see DfWarpView)
    unsigned long long seq, recorded_seq;            // sweeps prepared / recorded so far
    unsigned long long ring_seq[2];                  // the sweep whose completion ev_sweep_done[i] stands for (0: none)
    unsigned long long node_reader[2]; int nphase;  // [nphase] = the current node set's last reader, [nphase ^ 1] = the alternate's
    unsigned long long plan_reader[2];
    unsigned long long* plan_mask2[2]; unsigned int* plan_list2[2]; int pphase; unsigned hphase;
    unsigned long long* plan_code2[2];               // per strip item: which of its (patch, layer) cells read 4-bit codes
    uint16_t* pyr_mem; size_t pyr_cap;      // max-pyramid of the frame's dists image (warped sweep's depth cull), entries
    // scratch of dfusion_warp_solve_data_term (grown on demand)
    void* solver_ws; size_t solver_ws_cap;
    // points an indexed k-NN / warp pass left to the scan kernel: [0] count, [1..] ids
    int* pt_ids; size_t pt_ids_cap;
    int pt_image_cols;                                 // dfusion_warp_set_point_tiling: 0 = point queries in linear order
    // replica of the reference's nanoflann tree over the node positions (rebuilt by dfusion_warp_set_nodes)
    DfNfNode* nf_nodes; uint16_t* nf_vpos; size_t nf_nodes_cap, nf_vpos_cap; bool nf_ok; int nf_depth;
    // pipelined warped sweep: the launch plan (verdict masks of the strip items, the alive ones sorted by work), dfusion_warp.hip
    unsigned long long* plan_mask; unsigned int* plan_list; unsigned int* plan_hist; size_t plan_cap; int plan_phase;
    // per 8x8x8 block of the table planes (dfusion_warp_blocks.h; block grid blk_nbx x blk_nby x tab_zn / 8, whole table tiles):
    //   blk_state  0 = tables not built (DF_INDEX_TABLES_ON_DEMAND), 1 = built, 2 = built and a blend-model record written
    //   blk_wmax   max over the block's voxels of the weight sum (0 until built)
    //   blk_alive  this frame's verdicts ; blk_work [3][blk_cap] the frame's urgent-build / look-ahead-build / model work lists ;
    //   blk_cnt [2 sets][8] their lengths and the build passes' cursors (the sets alternate between passes: each pass zeroes the other one)
    // block blend models: entry-major [DF_BM_NU][blk_cap] node ids, {mid, half width} half pairs of the normalised and of the raw
    // weights (allocated with the first model), bm_cnt entry counts
    uint8_t* blk_state; float* blk_wmax; uint8_t* blk_alive; uint32_t* blk_work; uint32_t* blk_cnt; size_t blk_cap; int blk_phase;
    uint16_t* bm_idx; uint32_t* bm_lam; uint32_t* bm_w; uint8_t* bm_cnt; size_t bm_cap;
    // 4-bit neighbour codes (round 5; round 6: per 4 x 4 x 4 SUB-block, so that they exist at any node density): per voxel of a block
    // the model pass has visited, its k neighbours as positions in the union list of its sub-block (ascending node ids, <= 16).
    //   bm_ids   [block][64] u32: entry q * 16 + e = union entry e of column quadrant q (x >> 2 & 1 | (y >> 2 & 1) << 1), low half
    //            word for planes 0-3 of the block, high half word for planes 4-7 -- the dword a sweep lane loads to refill its wave's copies
    //   bm_coded [block] 1 = every sub-block's union fits 16 entries and the codes are written
    //   code_tab one u32 per voxel, tile-major like the tables but PATCH-major inside a tile plane
    uint32_t* code_tab; size_t code_cap; uint32_t* bm_ids; uint8_t* bm_coded;
    bool tab_complete;           // every block's tables are built
    int tab_sweeps;              // sweeps over the current tables so far (the models are made from the second one on)
    unsigned long long* dbg_swept;   // dfusion_warp_debug_counters: nullable device counter the sweeps through this handle add to
    bool alive_valid;            // blk_alive holds the verdicts of a sweep over the current tables (dfusion_warp_alive_blocks)
    // look-ahead work (tables / models of blocks a sweep does not need yet) runs on a side stream the handle owns, beside the sweep on
    // the caller's stream: ev_fork (caller's stream, after the verdict pass) releases it, ev_join (side stream, after its last kernel)
    // is waited for by the next call that touches the tables
    hipStream_t side; hipEvent_t ev_fork, ev_join; bool side_pending;
    // the verdict pass's list lengths, reported to pinned host memory by the plan kernel ([0] urgent builds, [1] models, [2] look-ahead
    // builds, [3] sweep number): read WITHOUT synchronisation by a later call -- a hint whether on-demand work is going on (then the side
    // stream is worth its fork / join, ~13 us per frame), never a condition of correctness
    volatile uint32_t* host_report;
};

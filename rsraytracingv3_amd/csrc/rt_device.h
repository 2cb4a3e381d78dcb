// rt_device.h — data shared between the C-ABI shim (host) and the gfx950 kernels.
//
// HBM layout of one context (DESIGN.md "Data layout"):
//   compiled scene   one allocation of float4 records re-laid out from the 112-byte AoS wire objects
//                    (objects.rs:202-214): see DeviceScene below
//   skybox           RGBA32F, row-major, row 0 = top (connection.rs:187-212)
//   random           R32F row-major (random_texture.rs:30-39), double buffered
//   rgba8            local_rows x W x 4 B: the Rgba8UnormSrgb target AND its own history
//   rgbf32 / hist    local_rows x W x 3 floats (optional output / float accumulation)
#pragma once

#include <cstdint>

#include "rt_math.h"
#include "rt_dirmask.h"

namespace rt355 {

constexpr uint32_t kTotalShards = 32;       // frame-total counters are sharded over this many 128 B lines
constexpr uint32_t kTotalShardStride = 16;  // u64 words between shards (128 B)
constexpr uint32_t kSphBatch = 8;  // sphere records consumed per scan-loop trip
constexpr uint32_t kMaxBatch = 64;  // frames one launch can render (rt_render_frames)
constexpr uint32_t kQueueShards = 32;  // path queue of the streamed shape: shards, each with its own head counters
constexpr uint32_t kQueueHeadWords = 2 * kQueueShards * 32;  // u32 words of head counters per launch (128 B apart)

// The compiled scene is ONE allocation of float4 records, laid out exactly as the kernels stage it into LDS
// ("scene image"); n = n_sph_pad, p = n_pl:
//   [0, n)            sph     {cx, cy, cz, r*r}, padded with never-hit records
//   [n, 2n)           smat0   {2*colour.rgb, roughness}            of the sphere in that slot
//   [2n, 3n)          smat1   {emission*strength .rgb, 1/(phong_alpha+1)}
//   [3n, 3n+p)        pl_a    {cx, cy, cz, half_size}
//   [3n+p, 3n+2p)     pl_b    {nx, ny, nz, bits(object index)}
//   [3n+2p, 3n+3p)    pmat0   as smat0, of the plane
//   [3n+3p, 3n+4p)    pmat1
//   [3n+4p, ...)      sph_obj original object index per sphere slot (ints, 4 per record; -1 in the padding)
// Materials are indexed by SLOT, so the winner's material needs no object index first (one dependent
// gather less per shaded hit), and for scenes of up to kImageLdsRecords records the whole image lives in LDS.
// Uniform grid over the spheres of a large scene (hundreds of spheres; built on the host at rt_set_scene,
// rt_grid.h).  Secondary rays walk it cell by cell (3-D DDA) and test only the spheres registered in the cells
// they cross instead of scanning every sphere.  cell_start[n_cells + 1] (u32) and items[] (u16 sphere slots) sit
// behind the scene image proper, at records off_cells / off_items; kernels stage them into LDS with the sphere
// records.  A sphere is registered in every cell its INFLATED ball touches, inflated enough that any ray the float
// test of ray.hlsl:53-68 can accept reaches the cell where the sphere is registered no later than the accepted
// distance (derivation at scan_grid in rt_kernels.hip) — for rays that start inside [near_min, near_max]; others
// scan linearly.
struct GridParams {
  uint32_t enabled;
  uint32_t nx, ny, nz, n_cells, n_items;
  uint32_t off_cells, off_items;  // float4-record offsets into the scene image
  float gmin[3], gmax[3];         // the grid's box
  float cell[3], inv_cell[3];     // cell size per axis and its reciprocal
  float near_min[3], near_max[3];
  float t_margin;                 // a hit ends the walk when it lies this far before the current cell's exit
};

struct DeviceScene {
  const float4 *image;
  GridParams grid;
  DirMaskParams dm;        // candidate-sphere masks for secondary rays (rt_dirmask.h): bit sets up to 128 spheres, candidate lists beyond
  uint32_t n_sph, n_pl, n_obj;
  uint32_t n_sph_pad;      // records in sph[]: n_sph rounded up to kSphBatch, plus one batch of read-ahead
  uint32_t image_records;  // float4 records in the whole image
  __host__ __device__ const float4 *sph() const { return image; }
  __host__ __device__ uint32_t off_smat0() const { return n_sph_pad; }
  __host__ __device__ uint32_t off_smat1() const { return 2u * n_sph_pad; }
  __host__ __device__ uint32_t off_pl_a() const { return 3u * n_sph_pad; }
  __host__ __device__ uint32_t off_pl_b() const { return 3u * n_sph_pad + n_pl; }
  __host__ __device__ uint32_t off_pmat0() const { return 3u * n_sph_pad + 2u * n_pl; }
  __host__ __device__ uint32_t off_pmat1() const { return 3u * n_sph_pad + 3u * n_pl; }
  __host__ __device__ uint32_t off_sph_obj() const { return 3u * n_sph_pad + 4u * n_pl; }
};
constexpr uint32_t kImageLdsRecords = 1024;  // 16 KB: scenes up to ~300 spheres are staged whole

// Defaults of the options that decide which kernels a launch runs and on what (rt_set_option); they live here, with the
// kernels, because profiles are tied to the sources that shape the measured launches (_native.kernel_source_hash).
constexpr int kDefaultDirmaskMinSpheres = 26;  // RT_OPT_DIRMASK_MIN
constexpr int kDefaultDirmaskCells = 4096;     // RT_OPT_DIRMASK_CELLS
constexpr int kDefaultDirmaskK = 12;           // RT_OPT_DIRMASK_K
constexpr int kDefaultDirlistCells = 16384;    // RT_OPT_DIRLIST_CELLS: cells aimed at by the candidate lists of large scenes
constexpr int kDefaultDirlistK = 10;           // RT_OPT_DIRLIST_K: their bins per face edge (A/B on one device at 1 024 spheres, reach 32: k = 6 / 8 / 10 / 12
                                               // 35.6 / 36.2 / 36.6 / 36.6 Grays/s; the table takes 164 / 292 / 456 / 657 MB)
constexpr int kDefaultDirlistReach = 32;       // RT_OPT_DIRLIST_REACH: the largest reach of an entry, in cells (16 / 32: 36.0 / 36.6)
constexpr int kDefaultTailSpillLists = 24;     // RT_OPT_TAIL_SPILL of scenes with candidate lists (one-wave workgroups: 16 / 24 / 32: 37.7 / 37.8 / 37.4)
constexpr int kDefaultGridMinSpheres = 129;    // RT_OPT_GRID_MIN
constexpr int kDefaultInflight = 4;            // RT_OPT_INFLIGHT
constexpr int kDefaultBatchInflight = 4;       // RT_OPT_BATCH_INFLIGHT = 0 (the library's rule): the next multi-frame launches render while this one's tail drains and
                                               // resolves; four while their sample buffers fit kBatchLanesBudget, else down to two
constexpr uint64_t kBatchLanesBudget = 2ull << 30;  // bytes of sample buffers the lanes of multi-frame launches may hold together under that rule
                                               // (round 3: 53.9 -> 57.8 Grays/s on the headline; one at a time: kernel times that do not overlap)
constexpr int kDefaultTailSpillSingle = 32;    // RT_OPT_TAIL_SPILL_SINGLE: single frames in flight of at least kTailMinTilesSingle tiles spill too.  Round 3 measured
                                               // it in runs of 20 frames between two synchronisations (-4 %: tail_stream's fixed ~35 us per launch) and left it
                                               // off; in an uninterrupted loop (blocks of 160 frames, A/B on one device, one frame per launch at 1080p) it is
                                               // 47.4 -> 49.1 Grays/s at 8 bounces
constexpr int kDefaultTailSpill = 32;          // RT_OPT_TAIL_SPILL (A/B on one device, thresholds 0 / 24 / 32 / 40: headline 50.3 / 52.9 / 52.9 /
                                               // 52.8 Grays/s, 80 slots 44.7 / 47.7 / 48.0 / 47.9, 3840x2160 54.2 / 58.5 / 58.9 / 58.9)

struct FrameParams {
  uint32_t width, height;      // full frame (Config.width/height, scene.rs:267-268)
  uint32_t reflection_limit;   // Config.reflection_limit
  uint32_t progressive_count;  // FrameData (connection.rs:13-16)
  // Frames of this launch.  A batch is rendered as one image of `batch` frames stacked on top of each
  // other, frame_tile_rows 8-row tile rows apart: the tile index decides the frame, the frame its
  // jitter.  batch == 1 is the plain frame; batch > 1 writes float samples (rgbf32 only, opacity 1)
  // that resolve_batch_kernel folds into the history in frame order.
  uint32_t batch, frame_tile_rows;
  float jitter[2 * kMaxBatch];  // {x, y} per frame
  float opacity, keep;         // 1/(progressive_count+1) and 1 - that (frag.hlsl:61-62)
  // pixel-independent part of create_ray (ray.hlsl:16-29), evaluated on the host
  f3 cam_pos, cam_right, cam_neg_up, top_left;
  float vp_width, vp_height;
  const float4 *hdri;
  uint32_t hdri_w, hdri_h;
  const float *rnd;
  uint32_t rnd_w, rnd_h;
  // float(size) and float(size - 1) of the two textures and float(width), float(height): the conversions the texel
  // addressing and the pixel coordinate need, done once on the host (set_size_floats) so that the kernels read them as
  // scalar operands
  float hdri_wf, hdri_hf, hdri_w1f, hdri_h1f, rnd_wf, rnd_hf, rnd_w1f, rnd_h1f, width_f, height_f;
  __host__ void set_size_floats() {
    hdri_wf = static_cast<float>(hdri_w);
    hdri_hf = static_cast<float>(hdri_h);
    hdri_w1f = static_cast<float>(hdri_w - 1u);
    hdri_h1f = static_cast<float>(hdri_h - 1u);
    rnd_wf = static_cast<float>(rnd_w);
    rnd_hf = static_cast<float>(rnd_h);
    rnd_w1f = static_cast<float>(rnd_w - 1u);
    rnd_h1f = static_cast<float>(rnd_h - 1u);
    width_f = static_cast<float>(width);
    height_f = static_cast<float>(height);
  }
  // row-strip partition (rt_set_partition)
  uint32_t rank, world, strip_rows, local_rows;
  // outputs
  uint32_t *rgba8;             // stored texels (written)
  const uint32_t *hist_rgba8;  // previous frame's texels (read when progressive_count > 0); may alias rgba8
  float *rgbf32;    // may be null
  float *hist_f32;  // RT_ACCUM_FLOAT32 history, else null
  uint32_t *segments;  // may be null
  int32_t *hits;       // may be null
  unsigned long long *counters;  // kTotalShards shards, kTotalShardStride words apart, each
                                 // {segments, rough hits, wave scan iterations (variant 2)}
  unsigned int *tile_counters;   // variant 2: 8 work counters, 32 words (128 B) apart, zero at launch
  const float *srgb_lut;         // 256 decoded values
  int accum_mode;
  // Primary-ray culling (variants 0, 1, 3): bit masks of the spheres whose inflated bound touches a
  // tile column's / tile row's slab of rays (cull_slabs kernel): [tiles_x column masks][tiles_y row
  // masks], mask_words 64-bit words each; a tile scans column & row.  null = scan every sphere.
  // cull_check != 0: scan both ways and count disagreements in counters[3].
  const unsigned long long *tile_masks;
  uint32_t mask_words;
  int cull_check;
  // Path queue of the streamed shape (variant 4): the primary pass appends the paths that go on after their first
  // segment, the bounce pass consumes them.  64 B of path state per record as four float4 planes (plane k at
  // queue + k * queue_plane), kQueueShards shards of queue_shard_cap records each; the shards' write heads are
  // tile_counters[s * 32], their read heads tile_counters[(kQueueShards + s) * 32] (zero at launch).  null = no queue.
  float4 *queue;
  uint32_t queue_shard_cap;
  size_t queue_plane;
  // Tail spill of the re-packing shape (variant 3, multi-frame launches; tail_stream in rt_kernels.hip): a workgroup whose
  // paths that hit number at most spill_max appends them — path state + hit, 80 B as five planes of the same queue — and
  // ends; tail_stream shades and continues them in full waves.  0 = off.  A workgroup spills once, so a shard of
  // ceil(workgroups / kQueueShards) * spill_max records cannot overflow (tail_spill_plan).
  uint32_t spill_max;
  // render_blocks' workgroup -> tile map (block_tile_pos in rt_kernels.hip): 0 = plain; otherwise 1 + the number of bands the
  // rotation starts at — the launch's number, so that consecutive launches give an XCD different bands of the frame
  uint32_t xcd_bands;
};

// resolve_batch_kernel: folds a batch's float samples into the history, frame by frame
struct ResolveParams {
  const float *samples;     // [batch][frame_stride pixels][3]
  uint32_t batch;
  uint32_t pixels;          // local_rows * width
  uint32_t frame_stride;    // pixels between frames in `samples`
  uint32_t count[kMaxBatch];  // progressive_count of each frame
  float opacity[kMaxBatch];   // 1 / (count + 1)
  uint32_t *rgba8;
  const uint32_t *hist_rgba8;
  float *rgbf32;
  float *hist_f32;
  const float *srgb_lut;
  int accum_mode;
  // intermediate frames (rt_keep_batch_frames): the stored texels of frame f go to kept[(f / keep_every) * pixels + p]
  // when (f + 1) % keep_every == 0; null = keep nothing
  uint32_t *kept;
  uint32_t keep_every;
};

// kernels (rt_kernels.hip)
struct LaunchConfig {
  int variant;            // -1 auto; 0 tiles, LDS geometry; 1 tiles, global-memory geometry;
                          // 2 resident waves with lane refill; 3 block-compacting tiles;
                          // 4 streamed: primary pass -> path queue -> resident waves refilled from the queue
  int refill_min;         // variant 2: idle lanes tolerated before a wave pulls new pixels
  int persistent_blocks;  // variant 2: grid size (resident workgroups)
  int tile_chunk;         // variant 2: consecutive 8x8 tiles per counter increment; variant 4: 64-path units a wave
                          // takes off the queue per counter increment
  int stream_blocks;      // variant 4: grid size of the bounce pass (resident workgroups)
  int tail_blocks;        // variant 3 with a tail spill (FrameParams::spill_max): grid size of tail_stream (resident workgroups)
  hipEvent_t tail_start;  // ... and an event to record between render_blocks and tail_stream (null: none)
  int flags;              // reserved for experiment switches
};
int persistent_blocks_per_cu(size_t lds_bytes);
bool has_ab_shapes();  // built with -DRT_AB_SHAPES: the resident-wave (2) and streamed (4) launch shapes exist
// event-count build only (make events): counts since the last reset — see RT_CNT in rt_kernels.hip; zeros otherwise
int prof_counts(unsigned long long out8[8], bool reset);
// The launch shape launch_render will use for this launch (the auto rule, and the fall-backs of a forced variant),
// and what the streamed shape needs: records per queue shard for this launch's tiles, resident workgroups per CU.
int pick_variant(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc);
uint32_t stream_queue_shard_cap(const FrameParams &fp);
int stream_blocks_per_cu(size_t lds_bytes);
// Tail spill (FrameParams::spill_max): the threshold this launch would run with when `wanted` is asked for (0: the launch
// shape, the scene or the queue budget rule it out) and the records per queue shard it needs (five float4 planes of
// kQueueShards shards); resident workgroups per CU of tail_stream.
constexpr size_t kTailQueueBudget = 1ull << 30;  // bytes of queue a launch may ask for; the threshold shrinks to fit (8 frames of 4K at the default threshold: 1.33 GB -> 24)
constexpr uint32_t kTailMinBouncesSingle = 7;    // ... and of at least this reflection limit: short paths leave too little tail to pay tail_stream's fixed cost (A/B on
                                                 // one device, one frame per launch at 1080p, spill off / on by bounce limit 2 / 3 / 5 / 8: 58.3 / 54.1, 56.6 / 52.6,
                                                 // 52.1 / 51.0, 47.4 / 48.8 Grays/s; 4K at 8: 48.7 / 51.4; the reference's 80 slots: 44.2 / 46.0)
constexpr uint32_t kTailMinBouncesSingleLists = 5;  // ... on a scene with candidate lists (1 024 spheres, spill off / on at 3 / 4 / 5 / 8 bounces: 38.5 / 35.1, 36.1 / 34.6,
                                                    // 33.1 / 34.0, 26.9 / 33.2; staged scenes at 6 / 7: 50.5 / 50.4, 48.8 / 49.5)
constexpr uint64_t kTailMinTilesSingle = 24000;  // ... of a single-frame launch in flight (a whole 1080p frame is 32 400; a rank's share of one is not worth it)
constexpr uint64_t kTailMinTiles = 90000;        // 8x8 tiles from which a multi-frame launch spills: three 1080p frames (97 200).  Round 3, one launch at a time: 2 frames
                                                 // per launch 42.8 Grays/s without / 41.0 with the spill, 4 frames 47.6 / 48.7, 8 frames 50.2 / 52.2.  With launches in flight
                                                 // (tools/small_batch_times.py, two / four in flight): 2 frames 45.0 / 46.9 and 49.1 / 48.4, 3 frames 44.9 / 49.1 and 50.7 / 50.8
constexpr uint64_t kTailMinTilesLists = 48000;   // ... on a scene with candidate lists, whose tails are longer: from two 1080p frames (2 frames 27.7 / 36.3 and 28.0 / 39.1 Grays/s,
                                                 // 3 frames 28.4 / 38.0 and 28.8 / 40.9; a single frame in flight: kTailMinTilesSingle)
uint32_t tail_spill_plan(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc, uint32_t wanted, uint32_t *shard_cap);
int tail_blocks_per_cu(size_t lds_bytes, bool lists);
int launch_sqrt_exhaustive(unsigned long long *d_mismatches, hipStream_t stream);
int launch_random_fill(float *dst, uint32_t count, unsigned long long seed, hipStream_t stream);
int launch_render(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc, hipStream_t stream,
                  int *variant_used = nullptr, int *stage_used = nullptr);  // the launch shape chosen (0..3) and what
                                                                            // it staged into LDS (render_tiles<STAGE>)
int launch_resolve(const ResolveParams &rp, hipStream_t stream, bool dry = false);  // dry: one workgroup (RT_OPT_FLAGS bit 4)
// rank 0 of a multi-GPU frame: interleave the row strips of all ranks (own buffer + the peers' buffers received
// into `stack`, rank r's rows at texel offset stack_offset[r]) into the height x width frame
constexpr uint32_t kMaxRanks = 64;
struct AssembleParams {
  uint32_t *frame;
  const uint32_t *own, *stack;
  uint32_t width, height, world, strip_rows;
  uint32_t stack_offset[kMaxRanks];  // in texels; [0] unused
};
int launch_assemble(const AssembleParams &ap, hipStream_t stream, bool dry = false);
int launch_cull_tiles(const DeviceScene &sc, const FrameParams &fp, unsigned long long *masks, hipStream_t stream);
// fills dm.table ([cells][bins] 64-bit sets) from the n registered spheres reg[] = {c, r_reg} (device memory)
int launch_dirmask_build(const DirMaskParams &dm, const float4 *reg, uint32_t n, unsigned long long *table, hipStream_t stream);
// fills the candidate-list table of a large scene (dm.lists; [cells][bins] 32-byte entries + the zero entry) from reg[]; *d_overflow
// (device memory, zeroed by the caller) counts the cells whose near spheres did not fit the builder's list
int launch_dirlist_build(const DirMaskParams &dm, const float4 *reg, uint32_t n, unsigned long long *table, uint32_t *d_overflow,
                         hipStream_t stream);
// dst = src, `bytes` bytes (both 16-byte aligned), by a copy kernel on `stream`
int launch_copy_texels(const void *src, void *dst, size_t bytes, hipStream_t stream);
int launch_math_unary(int op, const float *x, float *y, size_t n, hipStream_t stream);
int launch_math_binary(int op, const float *a, const float *b, float *y, size_t n, hipStream_t stream);

}  // namespace rt355

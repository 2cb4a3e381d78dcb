// rt_shim.hip — implementation of the rt355 C-ABI (include/rt355.h) on one gfx950 device.
//
// Stands in for src/gpu/{connection,render_target,random_texture}.rs of the reference: it
// owns the device copies of the bind-group resources (connection.rs:314-347), re-lays the
// wire-format scene out for the kernel, evaluates the per-frame uniform maths on the host
// with the same specified float32 functions the device uses, and launches the render kernel.
// No CPU rendering fallback exists here by design.
#include <hip/hip_runtime.h>
#include <roctracer/roctx.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/rt355.h"
#include "rt_device.h"
#include "rt_comm.h"
#include "rt_grid.h"

using namespace rt355;

struct rt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t upload_stream = nullptr;  // rt_set_random (other thread)
  bool owns_stream = true;
  // Per-frame HIP-event pairs and device counters live in rings of kRing slots so the
  // benchmark can read a whole timed region after one sync.  The ring is re-armed half a
  // ring ahead, so the most recent kRing/2 frames are always readable.
  static constexpr uint32_t kRing = 1024;
  static constexpr uint32_t kCounterWords = kTotalShards * kTotalShardStride;  // u64 words per ring slot
  static constexpr uint32_t kTileWords = kQueueHeadWords;  // u32 words per ring slot: variant 2's 8 work counters / variant 4's
                                                          // 2 x kQueueShards queue heads, 128 B apart
  std::vector<hipEvent_t> ev_start, ev_stop, ev_resolved;  // render kernel = start..stop, resolve pass = stop..resolved
  std::vector<hipEvent_t> ev_tail;                         // recorded between render_blocks and tail_stream (tail spill)
  std::vector<unsigned char> tail_ran;                     // the launch in that ring slot had a tail_stream
  bool timing_valid = false;

  // skybox
  float4 *d_hdri = nullptr;
  uint32_t hdri_w = 0, hdri_h = 0;
  // random texture: two device copies; `rnd_cur` is the one frames read
  std::mutex rnd_mutex;
  float *d_rnd[2] = {nullptr, nullptr};
  size_t rnd_cap[2] = {0, 0};
  uint32_t rnd_w[2] = {0, 0}, rnd_h[2] = {0, 0};
  hipEvent_t rnd_last_use[2] = {nullptr, nullptr};
  bool rnd_used[2] = {false, false};
  int rnd_cur = -1;
  // scene
  float4 *d_scene = nullptr;
  size_t scene_cap = 0;
  DeviceScene scene{};
  bool have_scene = false;
  rt_wire_config config{};
  // per-frame camera values (ray.hlsl:16-29)
  f3 cam_pos{}, cam_right{}, cam_neg_up{}, top_left{};
  float vp_width = 0, vp_height = 0;
  // target
  uint32_t width = 0, height = 0;
  uint32_t rank = 0, world = 1, strip_rows = 8, local_rows = 0;
  uint32_t out_flags = RT_OUT_RGBA8 | RT_OUT_RGBF32;
  int accum_mode = RT_ACCUM_SRGB8;
  uint32_t *d_rgba8 = nullptr;
  float *d_rgbf32 = nullptr;
  float *d_hist_f32 = nullptr;
  uint32_t *d_segments = nullptr;
  int32_t *d_hits = nullptr;
  uint32_t hits_limit = 0;  // reflection_limit the hits buffer was sized for
  uint32_t hits_rendered_limit = 0;  // reflection_limit of the frame the hits buffer holds
  bool hits_valid = false;           // a frame has been rendered into the hits buffer since it was allocated
  unsigned long long *d_counters = nullptr;  // kRing x kTotalShards x 128 B lines of {segments, rough hits, wave iterations}
  unsigned int *d_tile_counters = nullptr;   // kRing x kTileWords (variant 2 work counters)
  // caller-owned output buffers (rt_bind_outputs); null = the context's own
  uint32_t *ext_rgba8 = nullptr, *ext_hist_rgba8 = nullptr;
  float *ext_rgbf32 = nullptr;
  float *d_lut = nullptr;
  int variant = -1;         // RT_OPT_VARIANT (-1 = choose per launch)
  int refill_min = 1;       // RT_OPT_REFILL_MIN
  int blocks_per_cu = 0;    // RT_OPT_BLOCKS_PER_CU (0 = occupancy query)
  int cu_count = 0;
  int tile_chunk = 2;       // RT_OPT_TILE_CHUNK
  int tail_spill = -1;      // RT_OPT_TAIL_SPILL: a workgroup with at most this many paths left hands them to tail_stream; -1 = the default of
                            // the scene's kind (kDefaultTailSpill, kDefaultTailSpillLists)
  int tail_spill_single = kDefaultTailSpillSingle;  // RT_OPT_TAIL_SPILL_SINGLE: the same for single-frame launches in flight
  int tail_occupancy_blocks = 0;       // resident tail_stream workgroups per CU for tail_occupancy_records of scene image
  uint32_t tail_occupancy_records = 0;
  bool tail_occupancy_lists = false;
  int flags = 0;            // RT_OPT_FLAGS
  int timing = 1;           // RT_OPT_TIMING: 0 = single-frame launches in flight record no timing-only events
  std::vector<unsigned char> timed;  // the launch in that ring slot recorded its start event
  int cull = 1;             // RT_OPT_CULL: 0 off, 1 on, 2 on + self-check against the full scan
  unsigned long long *d_masks = nullptr;  // per-tile sphere masks of the primary-ray culling
  size_t masks_cap = 0;     // in 64-bit words
  bool masks_dirty = true;  // scene, size or partition changed since the masks were built
  float *d_samples = nullptr;  // staging of a batch's float samples (rt_render_frames), batch x padded rows x W x 3
  size_t samples_cap = 0;      // in floats
  // Several single-frame launches in flight (RT_OPT_INFLIGHT = n, default 4): frame k renders its float sample
  // on render stream k % n into pipe_samples[k % n] while the tails of the frames before it drain on the other
  // streams; the main stream waits for each render in order and folds the sample into the history (resolve), so
  // the history chain and everything the caller sees stay in frame order.
  // Multi-GPU frame (rt_comm_init): this context is rank comm_rank of comm_world; after every launch its strips
  // go to rank 0 over RCCL on comm_stream, beside the next launch.  Launch k renders into send[k & 1] with
  // send[(k & 1) ^ 1] as its history, so the gather of launch k reads a buffer launch k + 1 does not write.
  ncclComm_t comm = nullptr;
  uint32_t comm_rank = 0, comm_world = 0;
  // member of a single-process multi-device frame (rt_multi_create): the same alternating targets and rank-0 buffers as with
  // a communicator (comm_rank / comm_world give the layout), but the gather is driven by the rt_multi that owns the context
  // (peer copies to device 0) and gather_done[] are its events
  bool multi_member = false;
  bool exchanges() const { return comm != nullptr || multi_member; }
  hipStream_t comm_stream = nullptr;
  uint32_t *d_rgba8_alt = nullptr;  // send[1]; send[0] is d_rgba8
  hipEvent_t frame_ready[2] = {nullptr, nullptr}, gather_done[2] = {nullptr, nullptr};
  bool gather_pending[2] = {false, false};
  uint32_t *d_stack = nullptr, *d_frame = nullptr;  // rank 0: peers' strips as received; the assembled frame
  std::vector<uint32_t> rank_rows;                  // local rows of every rank
  uint64_t comm_launches = 0;
  int last_send = 0;  // send buffer the last launch wrote
  // intermediate frames of multi-frame launches (rt_keep_batch_frames)
  uint32_t keep_every = 0;
  uint32_t *d_kept = nullptr;
  size_t kept_cap = 0;       // in texels
  uint32_t kept_frames = 0;  // frames the last launch kept
  uint32_t kept_every_last = 0;
  int occupancy_blocks = 0;          // cached hipOccupancyMaxActiveBlocksPerMultiprocessor answer for ...
  uint32_t occupancy_records = ~0u;  // ... a scene image of this many records
  static constexpr int kMaxInflight = 8;
  int inflight = 0;  // RT_OPT_INFLIGHT; 0 = the library's rule (single_lanes in render_frames)
  int batch_inflight = 0;  // RT_OPT_BATCH_INFLIGHT; 0 = the library's rule (batch_lanes below)
  uint64_t prep_generation = 0;                 // bumped whenever prep_done is re-recorded
  uint64_t prep_seen[kMaxInflight] = {};        // generation each render stream has already waited for
  // direction masks (rt_dirmask.h): table of candidate-sphere sets for the secondary rays of scenes with up to 64 spheres
  int dirmask_on = 1;            // RT_OPT_DIRMASK
  int dirmask_min_spheres = kDefaultDirmaskMinSpheres;  // RT_OPT_DIRMASK_MIN: smaller scenes scan linearly (the lookup costs more than their scan: A/B on one
                                 // device, table / linear in Grays/s: 16 spheres 57.4 / 59.9, 24: 53.0 / 53.0, 28: 51.6 / 49.6, 32: 51.7 / 49.7,
                                 // 48: 49.2 / 44.0, 64: 46.9 / 38.7)
  int dirmask_cells = kDefaultDirmaskCells;      // RT_OPT_DIRMASK_CELLS: cells aimed at
  int dirmask_k = kDefaultDirmaskK;            // RT_OPT_DIRMASK_K: bins per cube-map face edge
  unsigned long long *d_dirmask = nullptr;
  size_t dirmask_cap = 0;        // in 64-bit words
  float4 *d_dirmask_reg = nullptr;
  std::vector<float4> dirmask_built_for;  // sphere records (+ one record of options) the table on the device was built from
  // candidate lists (rt_dirmask.h "Candidate lists"): scenes of more than 128 spheres
  int dirlist_on = 1;                               // RT_OPT_DIRLIST
  int dirlist_min_spheres = static_cast<int>(kDirMaskMaxSpheres) + 1;  // RT_OPT_DIRLIST_MIN
  int dirlist_cells = kDefaultDirlistCells;         // RT_OPT_DIRLIST_CELLS
  int dirlist_k = kDefaultDirlistK;                 // RT_OPT_DIRLIST_K
  int dirlist_reach = kDefaultDirlistReach;         // RT_OPT_DIRLIST_REACH
  size_t dirmask_reg_cap = 0;                       // float4 records d_dirmask_reg holds
  uint32_t *d_dirlist_overflow = nullptr;
  uint32_t dirlist_reach_used = 0;                  // the reach (cells) the current table was built with
  int grid_min_spheres = kDefaultGridMinSpheres;  // RT_OPT_GRID_MIN: scenes with at least this many spheres get the uniform grid; up to 128 spheres the
                               // direction-mask table serves (A/B on one device, tools/dirmask_sweep.sh, Grays/s table / grid / linear:
                               // 80 spheres 60.0 / 47.3 / 45.7, 96: 53.8 / 45.3 / 40.3, 128: 40.6 / 40.3 / 30.6 before the re-packing
                               // shape took 128-sphere scenes; round 2's linear / grid crossover was 88)
  hipStream_t render_stream[kMaxInflight] = {};
  hipEvent_t prep_done = nullptr;  // (the renders and resolves are joined on the per-launch timing events)
  bool resolve_pending[kMaxInflight] = {};  // pipe_samples[i] has been written; its resolve is ev_resolved[resolve_slot[i]]
  uint32_t resolve_slot[kMaxInflight] = {};
  bool prep_pending = false;                // work the renders depend on was queued on the main stream (prep_done)
  bool prep_pending_dirmask = false;        // the direction-mask build was queued on the main stream since the last launch
  float *d_pipe_samples[kMaxInflight] = {};
  size_t pipe_cap[kMaxInflight] = {};  // in floats
  // path queues of the streamed shape (variant 4): one per pipelined lane, the last one for multi-frame launches
  float4 *d_queue[kMaxInflight + 1] = {};
  size_t queue_cap[kMaxInflight + 1] = {};  // in float4 records (4 planes)
  int stream_occupancy_blocks = 0;
  uint32_t stream_occupancy_records = 0xffffffffu;
  uint64_t pipe_frames = 0;
  // Asynchronous presentation (rt_present_async / rt_present_wait): pinned host staging buffers the frame's texels are copied
  // into on a copy stream of their own, beside the next launches' rendering.
  static constexpr uint32_t kPresentSlots = RT355_PRESENT_SLOTS;
  hipStream_t present_stream = nullptr;
  hipEvent_t present_ready = nullptr;                  // the frame to present is complete (recorded on the stream that produced it)
  hipEvent_t present_done[kPresentSlots] = {};         // the copy into the slot has finished
  void *present_host[kPresentSlots] = {};
  size_t present_cap[kPresentSlots] = {};
  size_t present_bytes[kPresentSlots] = {};
  uint32_t present_rows[kPresentSlots] = {}, present_width[kPresentSlots] = {};
  bool present_queued[kPresentSlots] = {};             // a copy into the slot has been queued (and not been waited for since)
  uint32_t *d_present[kPresentSlots] = {};             // device-side snapshot of the frame the slot's copy reads: the target itself is
  size_t present_dev_cap[kPresentSlots] = {};          // the history, which the next launch overwrites in place
  uint32_t launch_seq = 0;                             // launches so far (FrameParams::xcd_bands)
  int present_guard = -1;                              // (unused since the snapshots: kept -1)
  rt_stats stats{};
  std::string error;
};

namespace {

// A named range around a phase of the shim, visible in rocprofv3 --marker-trace timelines next to the kernels it launches
// (a call into libroctx64 that does nothing while no profiler listens).
struct Range {
  explicit Range(const char *name) { (void)roctxRangePushA(name); }
  ~Range() { (void)roctxRangePop(); }
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};

int fail(rt_context *ctx, int code, const char *fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->error = buf;
  }
  return code;
}

#define RT_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE,          \
                  "%s failed: %s", #call, hipGetErrorString(e_));                                 \
  } while (0)

uint32_t count_local_rows(uint32_t height, uint32_t rank, uint32_t world, uint32_t strip) {
  uint32_t rows = 0;
  for (uint32_t y0 = 0, s = 0; y0 < height; y0 += strip, ++s) {
    if (s % world == rank) rows += (height - y0 < strip) ? height - y0 : strip;
  }
  return rows;
}

// Where rank r's rows sit in the concatenation of the peers' strips (ranks 1 .. world-1, each rank's local rows
// in order): the layout rank 0 receives into and assemble_strips_kernel reads.  Returns the peers' texel total.
size_t fill_assemble_layout(AssembleParams &ap, uint32_t width, uint32_t height, uint32_t world, uint32_t strip_rows) {
  ap.width = width;
  ap.height = height;
  ap.world = world;
  ap.strip_rows = strip_rows;
  size_t off = 0;
  for (uint32_t r = 1; r < world; ++r) {
    ap.stack_offset[r] = static_cast<uint32_t>(off);
    off += static_cast<size_t>(count_local_rows(height, r, world, strip_rows)) * width;
  }
  return off;
}

void free_target(rt_context *ctx) {
  (void)hipFree(ctx->d_rgba8);
  (void)hipFree(ctx->d_rgbf32);
  (void)hipFree(ctx->d_hist_f32);
  (void)hipFree(ctx->d_segments);
  (void)hipFree(ctx->d_hits);
  (void)hipFree(ctx->d_rgba8_alt);
  (void)hipFree(ctx->d_stack);
  (void)hipFree(ctx->d_frame);
  ctx->d_rgba8_alt = nullptr;
  ctx->d_stack = nullptr;
  ctx->d_frame = nullptr;
  ctx->gather_pending[0] = ctx->gather_pending[1] = false;
  ctx->d_rgba8 = nullptr;
  ctx->d_rgbf32 = nullptr;
  ctx->d_hist_f32 = nullptr;
  ctx->d_segments = nullptr;
  ctx->d_hits = nullptr;
  ctx->hits_limit = 0;
  ctx->hits_valid = false;
}

// (re)allocate per-pixel buffers for the current size / partition / outputs / accum mode
int alloc_target_buffers(rt_context *ctx, size_t px) {
  RT_HIP(ctx, hipMalloc(&ctx->d_rgba8, px * 4));
  RT_HIP(ctx, hipMemsetAsync(ctx->d_rgba8, 0, px * 4, ctx->stream));
  if (ctx->out_flags & RT_OUT_RGBF32) RT_HIP(ctx, hipMalloc(&ctx->d_rgbf32, px * 12));
  if (ctx->accum_mode == RT_ACCUM_FLOAT32) {
    RT_HIP(ctx, hipMalloc(&ctx->d_hist_f32, px * 12));
    RT_HIP(ctx, hipMemsetAsync(ctx->d_hist_f32, 0, px * 12, ctx->stream));
  }
  if (ctx->out_flags & RT_OUT_SEGMENTS) RT_HIP(ctx, hipMalloc(&ctx->d_segments, px * 4));
  if (ctx->exchanges()) {
    RT_HIP(ctx, hipMalloc(&ctx->d_rgba8_alt, px * 4));
    RT_HIP(ctx, hipMemsetAsync(ctx->d_rgba8_alt, 0, px * 4, ctx->stream));
    ctx->rank_rows.assign(ctx->comm_world, 0);
    size_t peer_px = 0;
    for (uint32_t r = 0; r < ctx->comm_world; ++r) {
      ctx->rank_rows[r] = count_local_rows(ctx->height, r, ctx->comm_world, ctx->strip_rows);
      if (r) peer_px += static_cast<size_t>(ctx->rank_rows[r]) * ctx->width;
    }
    if (ctx->comm_rank == 0) {
      if (peer_px) RT_HIP(ctx, hipMalloc(&ctx->d_stack, peer_px * 4));
      RT_HIP(ctx, hipMalloc(&ctx->d_frame, static_cast<size_t>(ctx->width) * ctx->height * 4));
      RT_HIP(ctx, hipMemsetAsync(ctx->d_frame, 0, static_cast<size_t>(ctx->width) * ctx->height * 4, ctx->stream));
    }
  }
  return RT_OK;
}

int alloc_target(rt_context *ctx) {
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
  if (ctx->present_stream) RT_HIP(ctx, hipStreamSynchronize(ctx->present_stream));  // a presentation copy may still read the old target
  ctx->present_guard = -1;
  free_target(ctx);
  // the path queues were sized for the old target's launches (a 4K or 64-frame launch may hold gigabytes): the next launch that
  // spills allocates what it needs
  for (hipStream_t rs : ctx->render_stream)
    if (rs) (void)hipStreamSynchronize(rs);
  for (int i = 0; i <= rt_context::kMaxInflight; ++i) {
    (void)hipFree(ctx->d_queue[i]);
    ctx->d_queue[i] = nullptr;
    ctx->queue_cap[i] = 0;
  }
  ctx->local_rows = count_local_rows(ctx->height, ctx->rank, ctx->world, ctx->strip_rows);
  ctx->masks_dirty = true;
  ctx->kept_frames = 0;
  const size_t px = static_cast<size_t>(ctx->local_rows) * ctx->width;
  if (px == 0) return RT_OK;
  const int rc = alloc_target_buffers(ctx, px);
  if (rc != RT_OK) {  // never leave a size behind that has no buffers: the context is back to "no target"
    free_target(ctx);
    ctx->width = ctx->height = ctx->local_rows = 0;
  }
  return rc;
}

int ensure_hits(rt_context *ctx, uint32_t limit) {
  if (!(ctx->out_flags & RT_OUT_HITS)) return RT_OK;
  if (ctx->d_hits && ctx->hits_limit >= limit) return RT_OK;
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_hits);
  ctx->d_hits = nullptr;
  ctx->hits_valid = false;
  const size_t n = static_cast<size_t>(ctx->local_rows) * ctx->width * (limit ? limit : 1);
  if (n) RT_HIP(ctx, hipMalloc(&ctx->d_hits, n * 4));
  ctx->hits_limit = limit;
  return RT_OK;
}

// ray.hlsl:16-29 — the part of create_ray that does not depend on the pixel.
void camera_setup(rt_context *ctx) {
  const rt_wire_config &c = ctx->config;
  const f3 position = mk3(c.position[0], c.position[1], c.position[2]);
  const f3 forward = mk3(c.forward[0], c.forward[1], c.forward[2]);
  const f3 right = mk3(c.right[0], c.right[1], c.right[2]);
  const f3 up = mk3(c.up[0], c.up[1], c.up[2]);
  const float fov_rad = (c.fov * kPi) * kInv180;
  const float half_width = tan_spec(fov_rad * 0.5f);
  const float half_height = (half_width * static_cast<float>(c.height)) / static_cast<float>(c.width);
  const f3 center = position - forward;
  const f3 left = fma3(right, -half_width, center);
  const f3 top = fma3(up, half_height, center);
  ctx->top_left = (left + top) - center;
  ctx->vp_width = half_width * 2.0f;
  ctx->vp_height = half_height * 2.0f;
  ctx->cam_pos = position;
  ctx->cam_right = right;
  ctx->cam_neg_up = -up;
}

// Copies local rows [first_row, first_row + n_rows) of a per-pixel output to the host (n_rows = UINT32_MAX: all).
template <typename T>
int read_back(rt_context *ctx, const T *src, void *dst, size_t bytes, size_t elem_bytes, const char *what,
              uint32_t first_row = 0, uint32_t n_rows = UINT32_MAX) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!dst) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "%s: dst is NULL", what);
  if (!src) return fail(ctx, RT_ERR_NOT_READY, "%s: output not enabled or no target allocated", what);
  if (n_rows == UINT32_MAX) n_rows = ctx->local_rows - (first_row <= ctx->local_rows ? first_row : ctx->local_rows);
  if (first_row > ctx->local_rows || n_rows > ctx->local_rows - first_row)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "%s: rows %u..%u of %u local rows", what, first_row, first_row + n_rows,
                ctx->local_rows);
  const size_t row_bytes = static_cast<size_t>(ctx->width) * elem_bytes;
  const size_t need = static_cast<size_t>(n_rows) * row_bytes;
  if (bytes != need) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "%s: %zu bytes given, %zu expected", what, bytes, need);
  if (need == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemcpyAsync(dst, reinterpret_cast<const unsigned char *>(src) + first_row * row_bytes, bytes,
                             hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

// The direction-mask table of the current spheres (rt_dirmask.h), rebuilt on the device only when the sphere records
// or the table's options changed (a moving camera re-sends the same objects every frame, connection.rs:373-385).
// Leaves ctx->scene.dm disabled when the scene is not served.
constexpr size_t kDirMaskMaxEntries = 8u << 20;  // 64 MB of one-word entries (128 MB with two words)
int build_dirmask(rt_context *ctx, const float4 *sph, uint32_t n) {
  float4 reg[kDirMaskMaxSpheres];
  uint32_t target = static_cast<uint32_t>(ctx->dirmask_cells);
  const bool take_outliers = !(ctx->flags & 32);  // RT_OPT_FLAGS bit 5: spheres far from the others stay in the table (A/B)
  DirMaskParams dm = dirmask_plan(sph, n, target, static_cast<uint32_t>(ctx->dirmask_k), reg, take_outliers);
  while (dm.enabled && static_cast<size_t>(dm.nx) * dm.ny * dm.nz * dm.bins > kDirMaskMaxEntries && target > 1u) {
    target /= 2u;
    dm = dirmask_plan(sph, n, target, static_cast<uint32_t>(ctx->dirmask_k), reg, take_outliers);
  }
  if (!dm.enabled) return RT_OK;
  const size_t entries = static_cast<size_t>(dm.nx) * dm.ny * dm.nz * dm.bins;
  if (entries > kDirMaskMaxEntries) return RT_OK;
  std::vector<float4> key(sph, sph + n);
  key.push_back(make_float4(static_cast<float>(dm.nx), static_cast<float>(dm.ny), static_cast<float>(dm.nz), static_cast<float>(dm.k)));
  key.push_back(make_float4(dm.cell[0], dm.bmin[0], dm.bmin[1], dm.bmin[2]));
  const bool same = ctx->d_dirmask && key.size() == ctx->dirmask_built_for.size() &&
                    std::memcmp(key.data(), ctx->dirmask_built_for.data(), key.size() * sizeof(float4)) == 0;
  if (!same) {
    ctx->dirmask_built_for.clear();
    const size_t need_words = (entries + 2u) * dm.words;  // + the all-zero entry and the outliers' entry
    if (ctx->dirmask_cap < need_words) {
      (void)hipFree(ctx->d_dirmask);
      ctx->d_dirmask = nullptr;
      ctx->dirmask_cap = 0;
      if (hipMalloc(&ctx->d_dirmask, need_words * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();  // no room for the table: the scene renders without one (linear scans), it does not fail
        ctx->d_dirmask = nullptr;
        return RT_OK;
      }
      ctx->dirmask_cap = need_words;
    }
    if (ctx->dirmask_reg_cap < kDirMaskMaxSpheres) {
      (void)hipFree(ctx->d_dirmask_reg);
      ctx->d_dirmask_reg = nullptr;
      ctx->dirmask_reg_cap = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_dirmask_reg, kDirMaskMaxSpheres * sizeof(float4)));
      ctx->dirmask_reg_cap = kDirMaskMaxSpheres;
    }
    RT_HIP(ctx, hipMemcpy(ctx->d_dirmask_reg, reg, n * sizeof(float4), hipMemcpyHostToDevice));
    if (launch_dirmask_build(dm, ctx->d_dirmask_reg, n, ctx->d_dirmask, ctx->stream) != 0)
      return fail(ctx, RT_ERR_DEVICE, "direction-mask build launch failed");
    ctx->dirmask_built_for = key;
    ctx->prep_pending_dirmask = true;
  }
  dm.table = ctx->d_dirmask;
  ctx->scene.dm = dm;
  return RT_OK;
}

// The candidate-list table of a large scene (rt_dirmask.h "Candidate lists"), built on the device and kept while the sphere
// records and the table's options stay the same.  Leaves ctx->scene.dm disabled when the scene is not served (no memory for the
// table: the caller then builds the sphere grid).  The caller has synchronised the context's stream.
constexpr size_t kDirListMaxBytes = 768ull << 20;
int build_dirlist(rt_context *ctx, const float4 *sph, uint32_t n) {
  const Range range("rt355 candidate lists");
  std::vector<float4> reg(n);
  uint32_t target = static_cast<uint32_t>(ctx->dirlist_cells);
  uint32_t reach = static_cast<uint32_t>(ctx->dirlist_reach);
  const uint32_t k = static_cast<uint32_t>(ctx->dirlist_k);
  auto bytes_of = [](const DirMaskParams &d) { return (static_cast<size_t>(d.nx) * d.ny * d.nz * d.bins + 1u) * 32u; };
  DirMaskParams dm = dirlist_plan(sph, n, target, k, reach, reg.data(), !(ctx->flags & 32));
  while (dm.enabled && bytes_of(dm) > kDirListMaxBytes && target > 1u) {
    target /= 2u;
    dm = dirlist_plan(sph, n, target, k, reach, reg.data(), !(ctx->flags & 32));
  }
  if (!dm.enabled || bytes_of(dm) > kDirListMaxBytes) return RT_OK;
  for (;;) {
    std::vector<float4> key(sph, sph + n);
    key.push_back(make_float4(static_cast<float>(dm.nx), static_cast<float>(dm.ny), static_cast<float>(dm.nz), static_cast<float>(dm.k)));
    key.push_back(make_float4(dm.cell[0], dm.bmin[0], dm.bmin[1], dm.bmin[2]));
    key.push_back(make_float4(dm.reach_unit, dm.s_cap, -1.0f, static_cast<float>(ctx->dirlist_reach)));  // (-1: never equal to a mask table's key)
    const bool same = ctx->d_dirmask && key.size() == ctx->dirmask_built_for.size() &&
                      std::memcmp(key.data(), ctx->dirmask_built_for.data(), key.size() * sizeof(float4)) == 0;
    if (same) {  // (the reach it ended up with is part of the kept state)
      dm = dirlist_plan(sph, n, target, k, ctx->dirlist_reach_used, reg.data(), !(ctx->flags & 32));
      break;
    }
    ctx->dirmask_built_for.clear();
    const size_t need_words = bytes_of(dm) / sizeof(unsigned long long);
    if (ctx->dirmask_cap < need_words) {
      (void)hipFree(ctx->d_dirmask);
      ctx->d_dirmask = nullptr;
      ctx->dirmask_cap = 0;
      if (hipMalloc(&ctx->d_dirmask, need_words * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();  // no room for the table: the scene renders without one (the sphere grid), it does not fail
        ctx->d_dirmask = nullptr;
        return RT_OK;
      }
      ctx->dirmask_cap = need_words;
    }
    if (ctx->dirmask_reg_cap < n) {
      (void)hipFree(ctx->d_dirmask_reg);
      ctx->d_dirmask_reg = nullptr;
      ctx->dirmask_reg_cap = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_dirmask_reg, static_cast<size_t>(n) * sizeof(float4)));
      ctx->dirmask_reg_cap = n;
    }
    if (!ctx->d_dirlist_overflow) RT_HIP(ctx, hipMalloc(&ctx->d_dirlist_overflow, sizeof(uint32_t)));
    uint32_t reach_now = reach;
    for (;;) {  // a cell whose near spheres overflow the builder's list asks for a shorter reach
      RT_HIP(ctx, hipMemcpy(ctx->d_dirmask_reg, reg.data(), static_cast<size_t>(n) * sizeof(float4), hipMemcpyHostToDevice));
      RT_HIP(ctx, hipMemsetAsync(ctx->d_dirlist_overflow, 0, sizeof(uint32_t), ctx->stream));
      if (launch_dirlist_build(dm, ctx->d_dirmask_reg, n, ctx->d_dirmask, ctx->d_dirlist_overflow, ctx->stream) != 0)
        return fail(ctx, RT_ERR_DEVICE, "candidate-list build launch failed");
      uint32_t overflow = 0;
      RT_HIP(ctx, hipMemcpyAsync(&overflow, ctx->d_dirlist_overflow, sizeof overflow, hipMemcpyDeviceToHost, ctx->stream));
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (overflow == 0u || reach_now <= 1u) break;
      reach_now = reach_now / 2u;
      dm = dirlist_plan(sph, n, target, k, reach_now, reg.data(), !(ctx->flags & 32));
    }
    ctx->dirlist_reach_used = reach_now;
    ctx->dirmask_built_for = key;
    ctx->prep_pending_dirmask = true;
    break;
  }
  dm.table = ctx->d_dirmask;
  ctx->scene.dm = dm;
  return RT_OK;
}

constexpr int kCommPatienceMs = 20000;  // how long leaving a communicator waits for the exchange in flight before it aborts it

}  // namespace

extern "C" {

static void release_comm(rt_context *ctx, int patience_ms, bool always_abort = false);

uint32_t rt_abi_version(void) { return RT355_ABI_VERSION; }
int rt_has_ab_shapes(void) { return has_ab_shapes() ? 1 : 0; }

int rt_create(int device_id, rt_context **out_ctx) {
  if (!out_ctx) return RT_ERR_INVALID_ARGUMENT;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return RT_ERR_NO_DEVICE;
  rt_context *ctx = new (std::nothrow) rt_context();
  if (!ctx) return RT_ERR_OUT_OF_MEMORY;
  ctx->device = device_id;
  auto bail = [&](int code) {
    rt_destroy(ctx);
    return code;
  };
  if (hipSetDevice(device_id) != hipSuccess) return bail(RT_ERR_NO_DEVICE);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return bail(RT_ERR_DEVICE);
  ctx->cu_count = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_DEVICE);
  if (hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_DEVICE);
  ctx->ev_start.assign(rt_context::kRing, nullptr);
  ctx->ev_stop.assign(rt_context::kRing, nullptr);
  ctx->ev_resolved.assign(rt_context::kRing, nullptr);
  ctx->ev_tail.assign(rt_context::kRing, nullptr);
  ctx->tail_ran.assign(rt_context::kRing, 0);
  ctx->timed.assign(rt_context::kRing, 0);
  for (uint32_t i = 0; i < rt_context::kRing; ++i)
    if (hipEventCreate(&ctx->ev_start[i]) != hipSuccess || hipEventCreate(&ctx->ev_stop[i]) != hipSuccess ||
        hipEventCreate(&ctx->ev_resolved[i]) != hipSuccess || hipEventCreate(&ctx->ev_tail[i]) != hipSuccess)
      return bail(RT_ERR_DEVICE);
  for (int i = 0; i < 2; ++i)
    if (hipEventCreateWithFlags(&ctx->rnd_last_use[i], hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_DEVICE);
  if (const char *pad = std::getenv("RT355_STREAM_PAD")) {  // experiment: shift the render streams' places in the runtime's round-robin over its hardware queues
    for (int i = 0; i < std::atoi(pad) && i < 8; ++i) {
      hipStream_t dummy;
      if (hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_DEVICE);
    }
  }
  for (int i = 0; i < rt_context::kMaxInflight; ++i)
    if (hipStreamCreateWithFlags(&ctx->render_stream[i], hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_DEVICE);
  if (hipEventCreateWithFlags(&ctx->prep_done, hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_DEVICE);
  if (hipMalloc(&ctx->d_counters, rt_context::kRing * rt_context::kCounterWords * sizeof(unsigned long long)) != hipSuccess)
    return bail(RT_ERR_OUT_OF_MEMORY);
  if (hipMemset(ctx->d_counters, 0, rt_context::kRing * rt_context::kCounterWords * sizeof(unsigned long long)) != hipSuccess)
    return bail(RT_ERR_DEVICE);
  if (hipMalloc(&ctx->d_tile_counters, rt_context::kRing * rt_context::kTileWords * sizeof(unsigned int)) != hipSuccess)
    return bail(RT_ERR_OUT_OF_MEMORY);
  if (hipMemset(ctx->d_tile_counters, 0, rt_context::kRing * rt_context::kTileWords * sizeof(unsigned int)) != hipSuccess)
    return bail(RT_ERR_DEVICE);
  if (hipMalloc(&ctx->d_lut, 256 * sizeof(float)) != hipSuccess) return bail(RT_ERR_OUT_OF_MEMORY);
  float lut[256];
  for (uint32_t b = 0; b < 256; ++b) lut[b] = srgb8_decode_spec(b);
  if (hipMemcpy(ctx->d_lut, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) return bail(RT_ERR_DEVICE);
  *out_ctx = ctx;
  return RT_OK;
}

int rt_destroy(rt_context *ctx) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(ctx->device);
  release_comm(ctx, kCommPatienceMs);       // before any synchronisation: a stuck exchange is aborted, not waited for
  (void)hipStreamSynchronize(ctx->stream);  // also when the caller bound the null stream (rt_set_stream)
  if (ctx->upload_stream) (void)hipStreamSynchronize(ctx->upload_stream);
  if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
  for (int i = 0; i < 2; ++i) {
    if (ctx->frame_ready[i]) (void)hipEventDestroy(ctx->frame_ready[i]);
    if (ctx->gather_done[i]) (void)hipEventDestroy(ctx->gather_done[i]);
  }
  if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
  free_target(ctx);
  (void)hipFree(ctx->d_hdri);
  (void)hipFree(ctx->d_rnd[0]);
  (void)hipFree(ctx->d_rnd[1]);
  (void)hipFree(ctx->d_scene);
  (void)hipFree(ctx->d_counters);
  (void)hipFree(ctx->d_tile_counters);
  (void)hipFree(ctx->d_masks);
  (void)hipFree(ctx->d_dirmask);
  (void)hipFree(ctx->d_dirmask_reg);
  (void)hipFree(ctx->d_dirlist_overflow);
  (void)hipFree(ctx->d_samples);
  (void)hipFree(ctx->d_queue[rt_context::kMaxInflight]);
  (void)hipFree(ctx->d_kept);
  for (int i = 0; i < rt_context::kMaxInflight; ++i) {
    if (ctx->render_stream[i]) (void)hipStreamSynchronize(ctx->render_stream[i]);
    (void)hipFree(ctx->d_pipe_samples[i]);
    (void)hipFree(ctx->d_queue[i]);
    if (ctx->render_stream[i]) (void)hipStreamDestroy(ctx->render_stream[i]);
  }
  if (ctx->prep_done) (void)hipEventDestroy(ctx->prep_done);
  if (ctx->present_stream) {
    (void)hipStreamSynchronize(ctx->present_stream);
    (void)hipStreamDestroy(ctx->present_stream);
  }
  if (ctx->present_ready) (void)hipEventDestroy(ctx->present_ready);
  for (uint32_t i = 0; i < rt_context::kPresentSlots; ++i) {
    if (ctx->present_done[i]) (void)hipEventDestroy(ctx->present_done[i]);
    if (ctx->present_host[i]) (void)hipHostFree(ctx->present_host[i]);
    (void)hipFree(ctx->d_present[i]);
  }
  (void)hipFree(ctx->d_lut);
  for (int i = 0; i < 2; ++i)
    if (ctx->rnd_last_use[i]) (void)hipEventDestroy(ctx->rnd_last_use[i]);
  for (hipEvent_t e : ctx->ev_start)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_stop)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_resolved)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_tail)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream && ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
  delete ctx;
  return RT_OK;
}

int rt_set_skybox(rt_context *ctx, uint32_t width, uint32_t height, const float *rgba32f) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!rgba32f || !width || !height) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_skybox: empty image");
  if (width > 65536u || height > 65536u || static_cast<uint64_t>(width) * height > (1ull << 31))
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_skybox: %ux%u is beyond 65536 per side / 2^31 texels", width, height);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t bytes = static_cast<size_t>(width) * height * 16;
  if (static_cast<size_t>(ctx->hdri_w) * ctx->hdri_h != static_cast<size_t>(width) * height) {
    (void)hipFree(ctx->d_hdri);
    ctx->d_hdri = nullptr;
    ctx->hdri_w = ctx->hdri_h = 0;
    RT_HIP(ctx, hipMalloc(&ctx->d_hdri, bytes));
  }
  RT_HIP(ctx, hipMemcpy(ctx->d_hdri, rgba32f, bytes, hipMemcpyHostToDevice));
  ctx->hdri_w = width;
  ctx->hdri_h = height;
  return RT_OK;
}

int rt_set_random(rt_context *ctx, uint32_t width, uint32_t height, const float *r32f) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!r32f || !width || !height) return RT_ERR_INVALID_ARGUMENT;  // no ctx->error write: other thread
  if (width > 65536u || height > 65536u || static_cast<uint64_t>(width) * height > (1ull << 31)) return RT_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(ctx->device) != hipSuccess) return RT_ERR_DEVICE;
  // Fill the copy no frame is reading, then publish it.  A frame launched meanwhile keeps
  // the old copy; one that used the target copy earlier is waited for on the upload stream.
  int target;
  bool wait_use;
  const size_t bytes = static_cast<size_t>(width) * height * 4;
  {
    // under the lock: a frame that chose `target` earlier has its last-use event recorded by now
    std::lock_guard<std::mutex> lk(ctx->rnd_mutex);
    target = (ctx->rnd_cur == 0) ? 1 : 0;
    wait_use = ctx->rnd_used[target];
    // every launch that read `target` has been joined to the context's stream by now (render_frames queues the join before it
    // releases the lock), and no later launch reads it before it is published below: an event recorded here is its last use
    if (wait_use && (hipEventRecord(ctx->rnd_last_use[target], ctx->stream) != hipSuccess ||
                     hipStreamWaitEvent(ctx->upload_stream, ctx->rnd_last_use[target], 0) != hipSuccess))
      return RT_ERR_DEVICE;
  }
  if (ctx->rnd_cap[target] < bytes) {
    if (hipStreamSynchronize(ctx->upload_stream) != hipSuccess) return RT_ERR_DEVICE;
    if (wait_use && hipEventSynchronize(ctx->rnd_last_use[target]) != hipSuccess) return RT_ERR_DEVICE;
    (void)hipFree(ctx->d_rnd[target]);
    ctx->d_rnd[target] = nullptr;
    ctx->rnd_cap[target] = 0;
    if (hipMalloc(&ctx->d_rnd[target], bytes) != hipSuccess) return RT_ERR_OUT_OF_MEMORY;
    ctx->rnd_cap[target] = bytes;
  }
  if (hipMemcpyAsync(ctx->d_rnd[target], r32f, bytes, hipMemcpyHostToDevice, ctx->upload_stream) != hipSuccess)
    return RT_ERR_DEVICE;
  if (hipStreamSynchronize(ctx->upload_stream) != hipSuccess) return RT_ERR_DEVICE;
  {
    std::lock_guard<std::mutex> lk(ctx->rnd_mutex);
    ctx->rnd_w[target] = width;
    ctx->rnd_h[target] = height;
    ctx->rnd_cur = target;
  }
  return RT_OK;
}

int rt_generate_random(rt_context *ctx, uint32_t width, uint32_t height, uint64_t seed) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!width || !height) return RT_ERR_INVALID_ARGUMENT;
  if (width > 65536u || height > 65536u || static_cast<uint64_t>(width) * height > (1ull << 31)) return RT_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(ctx->device) != hipSuccess) return RT_ERR_DEVICE;
  // same double-buffer protocol as rt_set_random, filled by a kernel instead of a host copy
  int target;
  bool wait_use;
  const size_t bytes = static_cast<size_t>(width) * height * 4;
  {
    // under the lock: a frame that chose `target` earlier has its last-use event recorded by now
    std::lock_guard<std::mutex> lk(ctx->rnd_mutex);
    target = (ctx->rnd_cur == 0) ? 1 : 0;
    wait_use = ctx->rnd_used[target];
    // every launch that read `target` has been joined to the context's stream by now (render_frames queues the join before it
    // releases the lock), and no later launch reads it before it is published below: an event recorded here is its last use
    if (wait_use && (hipEventRecord(ctx->rnd_last_use[target], ctx->stream) != hipSuccess ||
                     hipStreamWaitEvent(ctx->upload_stream, ctx->rnd_last_use[target], 0) != hipSuccess))
      return RT_ERR_DEVICE;
  }
  if (ctx->rnd_cap[target] < bytes) {
    if (hipStreamSynchronize(ctx->upload_stream) != hipSuccess) return RT_ERR_DEVICE;
    if (wait_use && hipEventSynchronize(ctx->rnd_last_use[target]) != hipSuccess) return RT_ERR_DEVICE;
    (void)hipFree(ctx->d_rnd[target]);
    ctx->d_rnd[target] = nullptr;
    ctx->rnd_cap[target] = 0;
    if (hipMalloc(&ctx->d_rnd[target], bytes) != hipSuccess) return RT_ERR_OUT_OF_MEMORY;
    ctx->rnd_cap[target] = bytes;
  }
  if (launch_random_fill(ctx->d_rnd[target], width * height, seed, ctx->upload_stream) != 0) return RT_ERR_DEVICE;
  if (hipStreamSynchronize(ctx->upload_stream) != hipSuccess) return RT_ERR_DEVICE;
  {
    std::lock_guard<std::mutex> lk(ctx->rnd_mutex);
    ctx->rnd_w[target] = width;
    ctx->rnd_h[target] = height;
    ctx->rnd_cur = target;
  }
  return RT_OK;
}

int rt_read_random(rt_context *ctx, float *dst, size_t bytes) {
  if (!ctx || !dst) return RT_ERR_INVALID_ARGUMENT;
  int cur;
  uint32_t w, h;
  {
    std::lock_guard<std::mutex> lk(ctx->rnd_mutex);
    cur = ctx->rnd_cur;
    w = cur >= 0 ? ctx->rnd_w[cur] : 0;
    h = cur >= 0 ? ctx->rnd_h[cur] : 0;
  }
  if (cur < 0) return fail(ctx, RT_ERR_NOT_READY, "rt_read_random: no random texture");
  if (bytes != static_cast<size_t>(w) * h * 4)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_random: %zu bytes given, %zu expected", bytes, static_cast<size_t>(w) * h * 4);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemcpy(dst, ctx->d_rnd[cur], bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_set_scene(rt_context *ctx, const void *objects, uint32_t object_count, const void *lights,
                 uint32_t light_count, const void *config112) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!config112 || (!objects && object_count)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_scene: NULL blob");
  const Range range("rt355 rt_set_scene");
  (void)lights;       // uploaded by the reference (connection.rs:379) but never read by the
  (void)light_count;  // shader (inputs.hlsl:67 has no use) — accepted and ignored
  RT_HIP(ctx, hipSetDevice(ctx->device));
  std::memcpy(&ctx->config, config112, sizeof ctx->config);
  if (!ctx->config.width || !ctx->config.height)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_scene: Config has a zero width or height");
  camera_setup(ctx);

  // Re-lay the AoS wire records out for the kernel: the scene image of rt_device.h.
  const auto *wire = static_cast<const uint8_t *>(objects);
  std::vector<float4> sph, smat0, smat1, pl_a, pl_b, pmat0, pmat1;
  std::vector<int> sph_obj;
  // Sphere records that repeat an earlier one bit for bit — the reference's buffer is 80 slots with a zero-filled tail (scene.rs:28,
  // bytes.rs:73-80): 15 spheres of radius 0 at the origin behind a scene of 64 — are left out of the image: the same arithmetic on the
  // same record gives the same distance, and the scan replaces its closest hit only by a strictly smaller one (ray.hlsl:98-106), in
  // ascending object order, so a repeat can never win against its first occurrence, whatever its material.  (Fifteen coincident
  // registered balls in the middle of the scene otherwise sit in the candidate set of every ray that passes the origin.)
  // RT_OPT_FLAGS bit 3 keeps them (A/B, tests).
  const bool drop_repeats = !(ctx->flags & 8);
  struct RecordKey {
    uint32_t w[4];
    bool operator==(const RecordKey &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
  };
  struct RecordHash {
    size_t operator()(const RecordKey &k) const {
      uint64_t h = 0x9e3779b97f4a7c15ull;
      for (uint32_t v : k.w) h = (h ^ v) * 0xff51afd7ed558ccdull + (h >> 29);
      return static_cast<size_t>(h);
    }
  };
  std::unordered_set<RecordKey, RecordHash> seen;
  uint32_t wire_spheres = 0;
  for (uint32_t i = 0; i < object_count; ++i) {
    rt_wire_object o;
    std::memcpy(&o, wire + static_cast<size_t>(i) * RT355_OBJECT_STRIDE, sizeof o);
    float inv_exp = 0.0f;
    if (o.roughness != 0.0f) {  // random.hlsl:30-33
      const float smoothness = 1.0f - o.roughness;
      const float phong_alpha = pow_spec(1000.0f, smoothness * smoothness);
      inv_exp = 1.0f / (phong_alpha + 1.0f);
    }
    const float4 m0 = make_float4(o.colour[0] * 2.0f, o.colour[1] * 2.0f, o.colour[2] * 2.0f, o.roughness);
    const float4 m1 = make_float4(o.emission[0] * o.emission_strength, o.emission[1] * o.emission_strength,
                                  o.emission[2] * o.emission_strength, inv_exp);
    if (o.option == 0u) {  // ray.hlsl:53-68
      wire_spheres += 1u;
      const float4 rec = make_float4(o.center[0], o.center[1], o.center[2], o.f32_data * o.f32_data);
      RecordKey key;
      std::memcpy(key.w, &rec, sizeof key.w);
      if (drop_repeats && !seen.insert(key).second) continue;
      sph.push_back(rec);
      sph_obj.push_back(static_cast<int>(i));
      smat0.push_back(m0);
      smat1.push_back(m1);
    } else if (o.option == 1u) {  // ray.hlsl:69-94
      pl_a.push_back(make_float4(o.center[0], o.center[1], o.center[2], o.f32_data));
      pl_b.push_back(make_float4(o.vec_data[0], o.vec_data[1], o.vec_data[2], u2f(i)));
      pmat0.push_back(m0);
      pmat1.push_back(m1);
    }  // any other option never intersects (ray.hlsl:53,69)
  }
  const size_t n_sph_live = sph.size(), n_pl = pl_a.size();
  // Pad the sphere records to whole scan batches plus one batch of read-ahead with records no
  // ray can accept (r*r = -inf makes the discriminant -inf), so the scan needs no tail code.
  const size_t n_sph = (n_sph_live + kSphBatch - 1) / kSphBatch * kSphBatch + kSphBatch;
  sph.resize(n_sph, make_float4(0.0f, 0.0f, 0.0f, -INFINITY));
  smat0.resize(n_sph, make_float4(0, 0, 0, 0));
  smat1.resize(n_sph, make_float4(0, 0, 0, 0));
  sph_obj.resize(n_sph, -1);
  const size_t sph_obj_rec = (n_sph + 3) / 4;  // ints packed into float4 records
  const size_t image_records = 3 * n_sph + 4 * n_pl + sph_obj_rec;
  // large scenes: candidate lists (built on the device, below); where those are off or cannot be built, a uniform grid over the
  // spheres, appended behind the image proper
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // frames in flight still read the old scene and its tables
  ctx->scene.dm = DirMaskParams{};
  if (ctx->dirmask_on && ctx->dirlist_on && n_sph_live >= static_cast<size_t>(ctx->dirlist_min_spheres) && n_sph_live <= kDirListMaxSpheres) {
    const int drc = build_dirlist(ctx, sph.data(), static_cast<uint32_t>(n_sph_live));
    if (drc != RT_OK) return drc;
  }
  const bool have_lists = ctx->scene.dm.enabled != 0u;
  GridBuild grid = have_lists ? GridBuild{} : build_grid(sph, n_sph_live, n_sph, static_cast<uint32_t>(ctx->grid_min_spheres));
  const size_t cells_rec = grid.params.enabled ? (grid.cell_start.size() * 4 + 15) / 16 : 0;
  const size_t items_rec = grid.params.enabled ? (grid.items.size() * 2 + 15) / 16 : 0;
  const size_t total = image_records + cells_rec + items_rec;
  std::vector<float4> blob(total, make_float4(0, 0, 0, 0));
  auto put = [&](size_t at, const void *src, size_t bytes) {
    if (bytes) std::memcpy(reinterpret_cast<unsigned char *>(blob.data()) + at * 16, src, bytes);
  };
  put(0, sph.data(), n_sph * 16);
  put(n_sph, smat0.data(), n_sph * 16);
  put(2 * n_sph, smat1.data(), n_sph * 16);
  put(3 * n_sph, pl_a.data(), n_pl * 16);
  put(3 * n_sph + n_pl, pl_b.data(), n_pl * 16);
  put(3 * n_sph + 2 * n_pl, pmat0.data(), n_pl * 16);
  put(3 * n_sph + 3 * n_pl, pmat1.data(), n_pl * 16);
  put(3 * n_sph + 4 * n_pl, sph_obj.data(), n_sph * 4);
  if (grid.params.enabled) {
    grid.params.off_cells = static_cast<uint32_t>(image_records);
    grid.params.off_items = static_cast<uint32_t>(image_records + cells_rec);
    put(grid.params.off_cells, grid.cell_start.data(), grid.cell_start.size() * 4);
    put(grid.params.off_items, grid.items.data(), grid.items.size() * 2);
  }

  if (ctx->scene_cap < total) {
    (void)hipFree(ctx->d_scene);
    ctx->d_scene = nullptr;
    ctx->scene_cap = 0;
    RT_HIP(ctx, hipMalloc(&ctx->d_scene, total * 16));
    ctx->scene_cap = total;
  }
  RT_HIP(ctx, hipMemcpy(ctx->d_scene, blob.data(), total * 16, hipMemcpyHostToDevice));
  ctx->scene.image = ctx->d_scene;
  ctx->scene.n_sph = static_cast<uint32_t>(n_sph_live);
  ctx->scene.n_sph_pad = static_cast<uint32_t>(n_sph);
  ctx->scene.n_pl = static_cast<uint32_t>(n_pl);
  ctx->scene.n_obj = object_count;
  ctx->scene.image_records = static_cast<uint32_t>(image_records);
  ctx->scene.grid = grid.params;
  if (!have_lists && ctx->dirmask_on && !grid.params.enabled && n_sph_live >= static_cast<size_t>(ctx->dirmask_min_spheres) &&
      n_sph_live <= kDirMaskMaxSpheres) {
    const int drc = build_dirmask(ctx, sph.data(), static_cast<uint32_t>(n_sph_live));
    if (drc != RT_OK) return drc;
  }
  if (!ctx->scene.dm.enabled && ctx->d_dirmask && ctx->dirmask_cap * sizeof(unsigned long long) > (64ull << 20)) {
    // a large table (candidate lists take hundreds of megabytes) of a scene that is gone, and this one uses none: give the memory back
    (void)hipFree(ctx->d_dirmask);
    ctx->d_dirmask = nullptr;
    ctx->dirmask_cap = 0;
    ctx->dirmask_built_for.clear();
  }
  ctx->have_scene = true;
  ctx->masks_dirty = true;
  ctx->stats.object_count = object_count;
  ctx->stats.sphere_count = wire_spheres;
  ctx->stats.spheres_scanned = ctx->scene.n_sph;
  ctx->stats.plane_count = ctx->scene.n_pl;
  return RT_OK;
}

int rt_resize(rt_context *ctx, uint32_t width, uint32_t height) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!width || !height) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_resize: zero size");
  // pixel indices are 32-bit in the kernels (with room for the 8-row padding of a multi-frame launch)
  if (width > 65536u || height > 65536u || static_cast<uint64_t>(width) * height > (1ull << 31))
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_resize: %ux%u is beyond 65536 per side / 2^31 pixels", width, height);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->width = width;
  ctx->height = height;
  return alloc_target(ctx);
}

int rt_set_partition(rt_context *ctx, uint32_t rank, uint32_t world, uint32_t strip_rows) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!world || rank >= world || !strip_rows)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_partition: rank %u / world %u / strip %u", rank, world, strip_rows);
  if (ctx->exchanges())
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_set_partition: the partition belongs to the communicator (rt_comm_init) / the rt_multi");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->rank = rank;
  ctx->world = world;
  ctx->strip_rows = strip_rows;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

// (the host-side twins of the scene structures — rt_grid_build_host, rt_dirmask_*_host, rt_dirlist_*_host — live in
// rt_host_twins.h: plain host code, also built on its own with the address and undefined-behaviour sanitizers, `make host-asan`)
#include "rt_host_twins.h"

int rt_comm_available(char *why_buf, size_t why_cap) {
  std::string why;
  const RcclApi &api = rccl_api(&why);
  if (why_buf && why_cap) {
    const std::string &text = api.ok() ? api.path : why;
    std::snprintf(why_buf, why_cap, "%s", text.c_str());
  }
  return api.ok() ? RT_OK : RT_ERR_NOT_READY;
}

int rt_comm_get_unique_id(void *id128) {
  if (!id128) return RT_ERR_INVALID_ARGUMENT;
  std::string why;
  const RcclApi &api = rccl_api(&why);
  if (!api.ok()) return RT_ERR_NOT_READY;
  ncclUniqueId id;
  if (api.GetUniqueId(&id) != ncclSuccess) return RT_ERR_DEVICE;
  static_assert(sizeof id == RT355_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof id);
  return RT_OK;
}

int rt_comm_init(rt_context *ctx, const void *id128, uint32_t rank, uint32_t world, uint32_t strip_rows) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!id128 || !world || rank >= world || !strip_rows || world > kMaxRanks)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_comm_init: rank %u / world %u (1..%u) / strip %u", rank, world, kMaxRanks, strip_rows);
  if (ctx->comm) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_comm_init: the context already has a communicator");
  if (ctx->multi_member) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_comm_init: the context belongs to an rt_multi");
  std::string why;
  const RcclApi &api = rccl_api(&why);
  if (!api.ok()) return fail(ctx, RT_ERR_NOT_READY, "rt_comm_init: no RCCL library could be loaded (%s)", why.c_str());
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = api.CommInitRank(&comm, static_cast<int>(world), id, static_cast<int>(rank));  // collective: all ranks
  if (rc != ncclSuccess) return fail(ctx, RT_ERR_DEVICE, "ncclCommInitRank (%s) failed: %s", api.path.c_str(), api.GetErrorString(rc));
  if (!ctx->comm_stream) {
    if (hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking) != hipSuccess) {
      (void)api.CommDestroy(comm);
      return fail(ctx, RT_ERR_DEVICE, "rt_comm_init: stream creation failed");
    }
    for (int i = 0; i < 2; ++i)
      if (hipEventCreateWithFlags(&ctx->frame_ready[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->gather_done[i], hipEventDisableTiming) != hipSuccess) {
        (void)api.CommDestroy(comm);
        return fail(ctx, RT_ERR_DEVICE, "rt_comm_init: event creation failed");
      }
  }
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  ctx->comm_launches = 0;
  ctx->rank = rank;
  ctx->world = world;
  ctx->strip_rows = strip_rows;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

// Leaving a communicator: waits for the exchange in flight — but only for `patience_ms`; an exchange that has not
// completed by then (a peer that never posted its side) is given up with ncclCommAbort, after which the RCCL kernel on
// the communication stream ends and the streams can be synchronised.  Never hangs.
// always_abort (rt_comm_abort): never ncclCommDestroy — that may wait for peers that never arrived (a hand-shake that failed on
// another rank) — whatever the streams look like.
static void release_comm(rt_context *ctx, int patience_ms, bool always_abort) {
  if (!ctx->comm) return;
  std::string why;
  const RcclApi &api = rccl_api(&why);
  bool done = false;
  for (int waited = 0; !done; waited += 2) {
    hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess && ctx->comm_stream) e = hipStreamQuery(ctx->comm_stream);
    (void)hipGetLastError();
    done = e != hipErrorNotReady;  // finished, or broken: either way nothing to wait for
    if (done || waited >= patience_ms) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  if (api.ok()) {
    if (done && !always_abort) (void)api.CommDestroy(ctx->comm);
    else (void)api.CommAbort(ctx->comm);
  }
  ctx->comm = nullptr;
  if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipGetLastError();
  ctx->gather_pending[0] = ctx->gather_pending[1] = false;
}

int rt_comm_destroy(rt_context *ctx) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!ctx->comm) return RT_OK;
  (void)hipSetDevice(ctx->device);
  release_comm(ctx, kCommPatienceMs);
  ctx->comm_world = 0;
  ctx->comm_rank = 0;
  ctx->rank = 0;
  ctx->world = 1;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

int rt_comm_idle(rt_context *ctx, int *idle) {
  if (!ctx || !idle) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipStreamQuery(ctx->stream);  // the renders of the internal streams are joined on this one
  if (e == hipSuccess && ctx->comm_stream) e = hipStreamQuery(ctx->comm_stream);
  if (e != hipSuccess && e != hipErrorNotReady) return fail(ctx, RT_ERR_DEVICE, "rt_comm_idle: %s", hipGetErrorString(e));
  (void)hipGetLastError();  // hipErrorNotReady is not an error
  *idle = e == hipSuccess ? 1 : 0;
  return RT_OK;
}

int rt_comm_abort(rt_context *ctx) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!ctx->comm) return RT_OK;
  (void)hipSetDevice(ctx->device);
  // no waiting first: the point is to get rid of an exchange that does not finish (ncclCommAbort is part of the
  // required symbol set of rt_comm.h, so the synchronisation inside release_comm cannot meet a live RCCL kernel)
  release_comm(ctx, 0, true);
  ctx->comm_world = 0;
  ctx->comm_rank = 0;
  ctx->rank = 0;
  ctx->world = 1;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

int rt_read_frame_rgba8(rt_context *ctx, void *dst, size_t bytes) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!dst) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_frame_rgba8: dst is NULL");
  if (!ctx->comm) return rt_read_rgba8(ctx, dst, bytes);  // one GPU: the local rows are the frame
  if (ctx->comm_rank != 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_frame_rgba8: the frame is assembled on rank 0 only");
  if (!ctx->d_frame) return fail(ctx, RT_ERR_NOT_READY, "rt_read_frame_rgba8: no target allocated");
  const size_t need = static_cast<size_t>(ctx->width) * ctx->height * 4;
  if (bytes != need) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_frame_rgba8: %zu bytes given, %zu expected", bytes, need);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_frame, bytes, hipMemcpyDeviceToHost, ctx->comm_stream));  // behind the last gather
  RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
  return RT_OK;
}

int rt_assemble_strips(rt_context *ctx, uint32_t width, uint32_t height, uint32_t world, uint32_t strip_rows, const void *rank0_rows,
                       const void *peer_rows, void *frame) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!width || !height || !world || world > kMaxRanks || !strip_rows || !rank0_rows || !frame || (world > 1 && !peer_rows))
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_assemble_strips: bad arguments");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  AssembleParams ap{};
  const size_t peer_px = fill_assemble_layout(ap, width, height, world, strip_rows);
  const size_t own_px = static_cast<size_t>(count_local_rows(height, 0, world, strip_rows)) * width;
  const size_t frame_px = static_cast<size_t>(width) * height;
  uint32_t *d_own = nullptr, *d_stack = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_own, own_px * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&d_stack, peer_px * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&d_out, frame_px * 4);
  if (e == hipSuccess) e = hipMemcpy(d_own, rank0_rows, own_px * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && peer_px) e = hipMemcpy(d_stack, peer_rows, peer_px * 4, hipMemcpyHostToDevice);
  int rc = 0;
  if (e == hipSuccess) {
    ap.frame = d_out;
    ap.own = d_own;
    ap.stack = d_stack;
    rc = launch_assemble(ap, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (rc == 0 && e == hipSuccess) e = hipMemcpy(frame, d_out, frame_px * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_own);
  (void)hipFree(d_stack);
  (void)hipFree(d_out);
  if (rc != 0 || e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "rt_assemble_strips failed: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_set_accum_mode(rt_context *ctx, int mode) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (mode != RT_ACCUM_SRGB8 && mode != RT_ACCUM_FLOAT32) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown accumulation mode %d", mode);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->accum_mode = mode;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

int rt_set_outputs(rt_context *ctx, uint32_t out_flags) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->out_flags = out_flags | RT_OUT_RGBA8;
  return (ctx->width && ctx->height) ? alloc_target(ctx) : RT_OK;
}

int rt_set_option(rt_context *ctx, int option, int value) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  switch (option) {
    case RT_OPT_VARIANT:
      if (value < -1 || value > 4) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown kernel variant %d", value);
      ctx->variant = value;
      return RT_OK;
    case RT_OPT_REFILL_MIN:
      ctx->refill_min = value;
      return RT_OK;
    case RT_OPT_BLOCKS_PER_CU:
      ctx->blocks_per_cu = value;
      return RT_OK;
    case RT_OPT_FLAGS:
      ctx->flags = value;
      return RT_OK;
    case RT_OPT_CULL:
      if (value < 0 || value > 2) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_CULL takes 0, 1 or 2");
      ctx->cull = value;
      return RT_OK;
    case RT_OPT_TILE_CHUNK:
      ctx->tile_chunk = value < 1 ? 1 : value;
      return RT_OK;
    case RT_OPT_GRID_MIN:
      ctx->grid_min_spheres = value < 1 ? 1 : value;
      return RT_OK;
    case RT_OPT_INFLIGHT:
      if (value < 0 || value > rt_context::kMaxInflight)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_INFLIGHT takes 0..%d", rt_context::kMaxInflight);
      ctx->inflight = value;  // a frame already in flight keeps its lane: lanes are only re-used after their resolve
      return RT_OK;
    case RT_OPT_DIRMASK:
      ctx->dirmask_on = value != 0;
      return RT_OK;
    case RT_OPT_DIRMASK_MIN:
      ctx->dirmask_min_spheres = value < 1 ? 1 : value;
      return RT_OK;
    case RT_OPT_DIRMASK_CELLS:
      if (value < 1 || value > 262144) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRMASK_CELLS takes 1..262144");
      ctx->dirmask_cells = value;
      return RT_OK;
    case RT_OPT_DIRMASK_K:
      if (value < 1 || value > static_cast<int>(kDirMaskMaxK)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRMASK_K takes 1..%u", kDirMaskMaxK);
      ctx->dirmask_k = value;
      return RT_OK;
    case RT_OPT_DIRLIST:
      ctx->dirlist_on = value != 0;
      return RT_OK;
    case RT_OPT_DIRLIST_MIN:
      ctx->dirlist_min_spheres = value < 1 ? 1 : value;
      return RT_OK;
    case RT_OPT_DIRLIST_CELLS:
      if (value < 1 || value > 2097152) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRLIST_CELLS takes 1..2097152");
      ctx->dirlist_cells = value;
      return RT_OK;
    case RT_OPT_DIRLIST_K:
      if (value < 1 || value > static_cast<int>(kDirMaskMaxK)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRLIST_K takes 1..%u", kDirMaskMaxK);
      ctx->dirlist_k = value;
      return RT_OK;
    case RT_OPT_DIRLIST_REACH:
      if (value < 1 || value > 128) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRLIST_REACH takes 1..128");
      ctx->dirlist_reach = value;
      return RT_OK;
    case RT_OPT_TIMING:
      ctx->timing = value != 0;
      return RT_OK;
    case RT_OPT_TAIL_SPILL:
      if (value < -1 || value > 128) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_TAIL_SPILL takes -1 (default) or 0..128");
      ctx->tail_spill = value;
      return RT_OK;
    case RT_OPT_TAIL_SPILL_SINGLE:
      if (value < 0 || value > 128) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_TAIL_SPILL_SINGLE takes 0..128");
      ctx->tail_spill_single = value;
      return RT_OK;
    case RT_OPT_BATCH_INFLIGHT:
      if (value < 0 || value > rt_context::kMaxInflight)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_BATCH_INFLIGHT takes 0..%d", rt_context::kMaxInflight);
      ctx->batch_inflight = value;
      return RT_OK;
    default:
      return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown option %d", option);
  }
}

int rt_set_variant(rt_context *ctx, int variant) { return rt_set_option(ctx, RT_OPT_VARIANT, variant); }

int rt_selftest_sqrt(rt_context *ctx, uint64_t *mismatches) {
  if (!ctx || !mismatches) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long *d = nullptr;
  RT_HIP(ctx, hipMalloc(&d, sizeof *d));
  hipError_t e = hipMemset(d, 0, sizeof *d);
  const int rc = launch_sqrt_exhaustive(d, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (rc != 0 || e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "sqrt self-test failed to run");
  *mismatches = h;
  return RT_OK;
}

// The exchange step of a tile-split frame: after the launch that wrote send buffer `b`, every peer
// sends its strips straight to rank 0 — grouped ncclSend / ncclRecv, one xGMI link per peer, no ring, no
// all-gather — on the communication stream, so it runs beside the next launch's rendering; rank 0 then
// interleaves the strips into the frame.
static int gather_frame(rt_context *ctx, int b) {
  const Range range("rt355 gather (RCCL)");
  std::string why;
  const RcclApi &api = rccl_api(&why);
  if (!api.ok()) return fail(ctx, RT_ERR_DEVICE, "RCCL is gone: %s", why.c_str());
  uint32_t *bufs[2] = {ctx->d_rgba8, ctx->d_rgba8_alt};
  RT_HIP(ctx, hipEventRecord(ctx->frame_ready[b], ctx->stream));
  RT_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->frame_ready[b], 0));
  const size_t row_bytes = static_cast<size_t>(ctx->width) * 4;
  ncclResult_t rc = ncclSuccess;
  if (ctx->comm_rank != 0) {
    if (ctx->local_rows) rc = api.Send(bufs[b], ctx->local_rows * row_bytes, ncclUint8, 0, ctx->comm, ctx->comm_stream);
  } else {
    AssembleParams ap{};
    fill_assemble_layout(ap, ctx->width, ctx->height, ctx->comm_world, ctx->strip_rows);
    rc = api.GroupStart();
    for (uint32_t r = 1; r < ctx->comm_world && rc == ncclSuccess; ++r)
      if (ctx->rank_rows[r])
        rc = api.Recv(ctx->d_stack + ap.stack_offset[r], ctx->rank_rows[r] * row_bytes, ncclUint8, static_cast<int>(r), ctx->comm,
                      ctx->comm_stream);
    const ncclResult_t rc2 = api.GroupEnd();
    if (rc == ncclSuccess) rc = rc2;
    if (rc == ncclSuccess) {
      ap.frame = ctx->d_frame;
      ap.own = bufs[b];
      ap.stack = ctx->d_stack;
      if (launch_assemble(ap, ctx->comm_stream, (ctx->flags & 16) != 0) != 0) return fail(ctx, RT_ERR_DEVICE, "assemble launch failed");
    }
  }
  if (rc != ncclSuccess) return fail(ctx, RT_ERR_DEVICE, "RCCL gather failed: %s", api.GetErrorString(rc));
  RT_HIP(ctx, hipEventRecord(ctx->gather_done[b], ctx->comm_stream));
  ctx->gather_pending[b] = true;
  return RT_OK;
}

// One launch of `batch` frames (1 = the plain rt_render): jitter {x, y} and progressive count per frame.
static int render_frames(rt_context *ctx, uint32_t batch, const float *jitter, const uint32_t *counts) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  const Range range(batch > 1u ? "rt355 rt_render_frames" : "rt355 rt_render");
  if (batch < 1u || batch > kMaxBatch)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frames: %u frames, 1..%u allowed per call", batch, kMaxBatch);
  if (batch > 1u && (ctx->out_flags & (RT_OUT_SEGMENTS | RT_OUT_HITS)))
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frames: the segments / hits outputs describe one frame; "
                                              "render batches with RT_OUT_RGBA8 | RT_OUT_RGBF32 only");
  if (static_cast<uint64_t>((ctx->local_rows + 7u) / 8u * 8u) * ctx->width * batch > 0xffffffffull)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frames: %u frames of %ux%u exceed 2^32 pixels per launch", batch,
                ctx->width, ctx->local_rows);
  const float jitter_x = jitter[0], jitter_y = jitter[1];
  const uint32_t progressive_count = counts[0];
  if (!ctx->d_hdri) return fail(ctx, RT_ERR_NOT_READY, "rt_render: no skybox (rt_set_skybox)");
  if (!ctx->have_scene) return fail(ctx, RT_ERR_NOT_READY, "rt_render: no scene (rt_set_scene)");
  if (!ctx->width) return fail(ctx, RT_ERR_NOT_READY, "rt_render: no target (rt_resize)");
  if (ctx->config.width != ctx->width || ctx->config.height != ctx->height)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render: Config is %ux%u but the target is %ux%u",
                ctx->config.width, ctx->config.height, ctx->width, ctx->height);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_hits(ctx, ctx->config.reflection_limit);
  if (rc != RT_OK) return rc;

  FrameParams fp{};
  // The random-texture copy this frame reads is chosen, the kernel launched and the copy's
  // "last use" event recorded under one lock: rt_set_random (other thread) waits on that event
  // before it refills the copy, so it must never observe the choice without the event.
  std::unique_lock<std::mutex> rnd_lock(ctx->rnd_mutex);
  const int rnd = ctx->rnd_cur;
  if (rnd >= 0) {
    fp.rnd = ctx->d_rnd[rnd];
    fp.rnd_w = ctx->rnd_w[rnd];
    fp.rnd_h = ctx->rnd_h[rnd];
  }
  if (rnd < 0) return fail(ctx, RT_ERR_NOT_READY, "rt_render: no random texture (rt_set_random)");
  fp.width = ctx->width;
  fp.height = ctx->height;
  fp.reflection_limit = ctx->config.reflection_limit;
  fp.progressive_count = progressive_count;
  fp.batch = batch;
  fp.frame_tile_rows = (ctx->local_rows + 7u) / 8u;
#ifndef RT_XCD_BANDS_SINGLE
#define RT_XCD_BANDS_SINGLE 0
#endif
  fp.xcd_bands = (batch > 1u || RT_XCD_BANDS_SINGLE) ? 1u + (ctx->launch_seq++ & 7u) : 0u;
  bool jitter_small = true;
  for (uint32_t f = 0; f < batch; ++f) {
    fp.jitter[2 * f] = jitter[2 * f];
    fp.jitter[2 * f + 1] = jitter[2 * f + 1];
    jitter_small = jitter_small && fabsf(jitter[2 * f]) <= 0.5f && fabsf(jitter[2 * f + 1]) <= 0.5f;
  }
  (void)jitter_x;
  (void)jitter_y;
  fp.opacity = 1.0f / static_cast<float>(progressive_count + 1u);  // frag.hlsl:61
  fp.keep = 1.0f - fp.opacity;
  fp.cam_pos = ctx->cam_pos;
  fp.cam_right = ctx->cam_right;
  fp.cam_neg_up = ctx->cam_neg_up;
  fp.top_left = ctx->top_left;
  fp.vp_width = ctx->vp_width;
  fp.vp_height = ctx->vp_height;
  fp.hdri = ctx->d_hdri;
  fp.hdri_w = ctx->hdri_w;
  fp.hdri_h = ctx->hdri_h;
  fp.set_size_floats();
  fp.rank = ctx->rank;
  fp.world = ctx->world;
  fp.strip_rows = ctx->strip_rows;
  fp.local_rows = ctx->local_rows;
  fp.rgba8 = ctx->ext_rgba8 ? ctx->ext_rgba8 : ctx->d_rgba8;
  fp.hist_rgba8 = ctx->ext_rgba8 ? (ctx->ext_hist_rgba8 ? ctx->ext_hist_rgba8 : ctx->ext_rgba8) : ctx->d_rgba8;
  fp.rgbf32 = ctx->ext_rgba8 ? ctx->ext_rgbf32 : ctx->d_rgbf32;
  const int send = static_cast<int>(ctx->comm_launches & 1u);
  if (ctx->exchanges()) {  // alternate targets: the gather of this launch must not meet the next launch's writes
    if (ctx->ext_rgba8) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render: rt_bind_outputs and a multi-GPU frame exclude each other");
    uint32_t *bufs[2] = {ctx->d_rgba8, ctx->d_rgba8_alt};
    fp.rgba8 = bufs[send];
    fp.hist_rgba8 = bufs[send ^ 1];
    // the gather that read this target two launches ago must be over before anything writes it again
    if (ctx->gather_pending[send]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gather_done[send], 0));
  }
  fp.hist_f32 = ctx->d_hist_f32;
  fp.segments = ctx->d_segments;
  fp.hits = ctx->d_hits;
  const uint32_t slot = static_cast<uint32_t>(ctx->stats.frames % rt_context::kRing);
  fp.counters = ctx->d_counters + rt_context::kCounterWords * static_cast<size_t>(slot);
  fp.tile_counters = ctx->d_tile_counters + rt_context::kTileWords * static_cast<size_t>(slot);
  fp.srgb_lut = ctx->d_lut;
  fp.accum_mode = ctx->accum_mode;
  // A batch renders float samples of all its frames (each as a frame of its own: count 0, opacity 1)
  // into the staging buffer; resolve_batch_kernel then replays the history update frame by frame.
  ResolveParams rp{};
  // Single frames keep two launches in flight (see rt_context::inflight) unless a per-frame diagnostic output
  // (segments, hits) is on: those buffers describe ONE frame and two kernels must not write them at once.
  // Multi-frame launches overlap the same way when RT_OPT_BATCH_INFLIGHT >= 2 (each lane then stages a whole batch).
  // Multi-frame launches in flight when the host left the choice to the library: four (8 frames of 1080p per launch, 2 -> 4 in flight:
  // headline +0.7 %, the reference's 80 slots +1.8 %, 1 024 spheres +1.4 %; 4 frames per launch +5 %; tools/small_batch_times.py) while
  // the lanes' sample buffers stay within kBatchLanesBudget together, never fewer than two (8 frames of 3840x2160 are 0.8 GB a lane: two).
  int batch_lanes = ctx->batch_inflight;
  if (batch > 1u && batch_lanes == 0) {
    const uint64_t lane_bytes = static_cast<uint64_t>((ctx->local_rows + 7u) / 8u * 8u) * ctx->width * batch * 3u * sizeof(float);
    const uint64_t fit = kBatchLanesBudget / (lane_bytes ? lane_bytes : 1u);
    batch_lanes = fit >= static_cast<uint64_t>(kDefaultBatchInflight) ? kDefaultBatchInflight : (fit < 2u ? 2 : static_cast<int>(fit));
  }
  // Single frames in flight when the host left the choice to the library: kDefaultInflight (four), five for short launches — paths of one or two segments
  // (tools/inflight_by_bounces.py, 1080p, 4 / 5 in flight: one bounce 0.0489 / 0.0449 ms per frame, two 0.0688 / 0.0676, three to six level,
  // eight and more 0.1162 / 0.1146 .. 0.1178 — five is bimodal there, and cost bench.py's one-frame leg 4 % in one run of three).
  // A rank's share of a tile-split frame (fewer tiles than a whole 1080p frame: the launches that do not spill either) gains the same way
  // (tools/inflight_times.py, 4 / 5 in flight: 1080p N = 2 / 4 / 8 0.0651 / 0.0637, 0.0380 / 0.0352, 0.0287 / 0.0242 ms; 3840x2160 N = 8 0.0625 / 0.0592).
  const uint64_t single_tiles = static_cast<uint64_t>((ctx->width + 7u) / 8u) * ((ctx->local_rows + 7u) / 8u);
  const bool short_launch = ctx->config.reflection_limit <= 2u || single_tiles < kTailMinTilesSingle;
  const int single_lanes = ctx->inflight != 0 ? ctx->inflight : (short_launch ? kDefaultInflight + 1 : kDefaultInflight);
  const int lanes = batch == 1u ? single_lanes : batch_lanes;
  const bool pipelined = lanes >= 2 && !(ctx->out_flags & (RT_OUT_SEGMENTS | RT_OUT_HITS));
  const int lane = static_cast<int>(ctx->pipe_frames % static_cast<uint64_t>(lanes < 1 ? 1 : lanes));
  const int stream_index = lane;
  hipStream_t render_stream = pipelined ? ctx->render_stream[stream_index] : ctx->stream;
  float *staging = nullptr;
  if (batch > 1u || pipelined) {
    const size_t frame_stride = static_cast<size_t>(fp.frame_tile_rows) * 8u * ctx->width;
    const size_t need = frame_stride * batch * 3u;
    float **buf = pipelined ? &ctx->d_pipe_samples[lane] : &ctx->d_samples;
    size_t *cap = pipelined ? &ctx->pipe_cap[lane] : &ctx->samples_cap;
    if (*cap < need) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the main stream has waited for every render so far
      (void)hipFree(*buf);
      *buf = nullptr;
      *cap = 0;
      RT_HIP(ctx, hipMalloc(buf, need * sizeof(float)));
      *cap = need;
      if (pipelined) ctx->resolve_pending[lane] = false;
    }
    staging = *buf;
    rp.samples = staging;
    rp.batch = batch;
    rp.pixels = ctx->local_rows * ctx->width;
    rp.frame_stride = static_cast<uint32_t>(frame_stride);
    for (uint32_t f = 0; f < batch; ++f) {
      rp.count[f] = counts[f];
      rp.opacity[f] = 1.0f / static_cast<float>(counts[f] + 1u);  // frag.hlsl:61
    }
    rp.rgba8 = fp.rgba8;
    rp.hist_rgba8 = fp.hist_rgba8;
    rp.rgbf32 = fp.rgbf32;
    rp.hist_f32 = fp.hist_f32;
    rp.srgb_lut = fp.srgb_lut;
    rp.accum_mode = fp.accum_mode;
    rp.kept = nullptr;
    rp.keep_every = 1;
    ctx->kept_frames = 0;
    if (ctx->keep_every != 0u && batch > 1u && batch / ctx->keep_every != 0u) {
      const uint32_t n_kept = batch / ctx->keep_every;
      const size_t need_kept = static_cast<size_t>(n_kept) * rp.pixels;
      if (ctx->kept_cap < need_kept) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_kept);
        ctx->d_kept = nullptr;
        ctx->kept_cap = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_kept, need_kept * 4));
        ctx->kept_cap = need_kept;
      }
      rp.kept = ctx->d_kept;
      rp.keep_every = ctx->keep_every;
      ctx->kept_frames = n_kept;
      ctx->kept_every_last = ctx->keep_every;
    }
    fp.progressive_count = 0;
    fp.opacity = 1.0f;
    fp.keep = 0.0f;
    fp.rgba8 = nullptr;
    fp.hist_rgba8 = nullptr;
    fp.rgbf32 = staging;
    fp.hist_f32 = nullptr;
    fp.accum_mode = 0;
  }

  // Primary-ray culling masks: rebuilt when the scene, the size or the partition changed; valid for
  // every jitter below half a pixel (larger jitters render without culling).
  bool prep_queued = false;  // something the render kernel reads was queued on the main stream just now
  fp.tile_masks = nullptr;
  fp.mask_words = 0;
  fp.cull_check = 0;
  if (ctx->cull != 0 && ctx->scene.n_sph != 0 && jitter_small) {
    const uint32_t words = (ctx->scene.n_sph + 63u) / 64u;
    const size_t tiles = static_cast<size_t>((ctx->width + 7) / 8) + ((ctx->local_rows + 7) / 8);  // column + row slabs
    if (ctx->masks_cap < tiles * words) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->d_masks);
      ctx->d_masks = nullptr;
      ctx->masks_cap = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_masks, tiles * words * sizeof(unsigned long long)));
      ctx->masks_cap = tiles * words;
      ctx->masks_dirty = true;
    }
    fp.mask_words = words;
    if (ctx->masks_dirty) {
      if (launch_cull_tiles(ctx->scene, fp, ctx->d_masks, ctx->stream) != 0)
        return fail(ctx, RT_ERR_DEVICE, "cull_tiles launch failed");
      ctx->masks_dirty = false;
      prep_queued = true;
    }
    fp.tile_masks = ctx->d_masks;
    fp.cull_check = ctx->cull == 2;
  }
  if (slot % (rt_context::kRing / 2) == 0 && ctx->stats.frames != 0) {  // re-arm the half we are entering
    RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_counters + rt_context::kTileWords * static_cast<size_t>(slot), 0,
                               (rt_context::kRing / 2) * rt_context::kTileWords * sizeof(unsigned int), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(ctx->d_counters + rt_context::kCounterWords * static_cast<size_t>(slot), 0,
                               (rt_context::kRing / 2) * rt_context::kCounterWords * sizeof(unsigned long long), ctx->stream));
    prep_queued = true;
  }
  if (ctx->prep_pending_dirmask) {  // rt_set_scene queued the direction-mask build on the main stream
    prep_queued = true;
    ctx->prep_pending_dirmask = false;
  }
  if (prep_queued) {  // render streams must not start before the masks / re-armed counters are there
    RT_HIP(ctx, hipEventRecord(ctx->prep_done, ctx->stream));
    ctx->prep_pending = true;
    ctx->prep_generation += 1;
  }
  if (pipelined) {
    // this lane's sample buffer is free once the resolve that read it (two frames ago) has run
    if (ctx->resolve_pending[lane]) RT_HIP(ctx, hipStreamWaitEvent(render_stream, ctx->ev_resolved[ctx->resolve_slot[lane]], 0));
    if (ctx->prep_pending && ctx->prep_seen[stream_index] != ctx->prep_generation) {
      RT_HIP(ctx, hipStreamWaitEvent(render_stream, ctx->prep_done, 0));
      ctx->prep_seen[stream_index] = ctx->prep_generation;
    }
  }
  LaunchConfig lc{};
  lc.variant = ctx->variant;
  lc.refill_min = ctx->refill_min < 1 ? 1 : (ctx->refill_min > 64 ? 64 : ctx->refill_min);
  lc.tile_chunk = ctx->tile_chunk;
  lc.flags = ctx->flags;
  fp.queue = nullptr;
  fp.queue_shard_cap = 0;
  fp.queue_plane = 0;
  if (pick_variant(ctx->scene, fp, lc) == 4) {  // streamed: the launch needs a path queue and the bounce pass its grid
    const int qi = pipelined ? lane : static_cast<int>(rt_context::kMaxInflight);
    const uint32_t shard_cap = stream_queue_shard_cap(fp);
    const size_t plane = static_cast<size_t>(shard_cap) * kQueueShards;
    if (ctx->queue_cap[qi] < 4 * plane) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the main stream has waited for every render so far
      (void)hipFree(ctx->d_queue[qi]);
      ctx->d_queue[qi] = nullptr;
      ctx->queue_cap[qi] = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_queue[qi], 4 * plane * sizeof(float4)));
      ctx->queue_cap[qi] = 4 * plane;
    }
    fp.queue = ctx->d_queue[qi];
    fp.queue_shard_cap = shard_cap;
    fp.queue_plane = plane;
    if (ctx->stream_occupancy_records != ctx->scene.image_records) {
      ctx->stream_occupancy_blocks = stream_blocks_per_cu(static_cast<size_t>(ctx->scene.image_records) * sizeof(float4));
      ctx->stream_occupancy_records = ctx->scene.image_records;
    }
    const int per_cu_stream = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (ctx->stream_occupancy_blocks > 0 ? ctx->stream_occupancy_blocks : 4);
    lc.stream_blocks = per_cu_stream * ctx->cu_count;
  }
  fp.spill_max = 0;
  const int spill_default = (ctx->scene.dm.enabled && ctx->scene.dm.lists) ? kDefaultTailSpillLists : kDefaultTailSpill;
  const int spill_wanted = batch > 1u ? (ctx->tail_spill >= 0 ? ctx->tail_spill : spill_default) : ctx->tail_spill_single;
  if (spill_wanted > 0 && fp.queue == nullptr) {  // tail spill of the re-packing shape: needs the queue and tail_stream's grid
    uint32_t shard_cap = 0;
    const uint32_t t = tail_spill_plan(ctx->scene, fp, lc, static_cast<uint32_t>(spill_wanted), &shard_cap);
    if (t != 0u) {
      const int qi = pipelined ? lane : static_cast<int>(rt_context::kMaxInflight);
      const size_t plane = static_cast<size_t>(shard_cap) * kQueueShards;
      bool have = ctx->queue_cap[qi] >= 5 * plane;
      if (!have) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the main stream has waited for every render so far
        (void)hipFree(ctx->d_queue[qi]);
        ctx->d_queue[qi] = nullptr;
        ctx->queue_cap[qi] = 0;
        if (hipMalloc(&ctx->d_queue[qi], 5 * plane * sizeof(float4)) == hipSuccess) {
          ctx->queue_cap[qi] = 5 * plane;
          have = true;
          ctx->stats.tail_spill_unavailable = 0;
        } else {
          (void)hipGetLastError();  // no memory for the queue: the launch runs without the spill — and says so (rt_stats.tail_spill_unavailable)
          ctx->d_queue[qi] = nullptr;
          ctx->stats.tail_spill_unavailable = 1;
        }
      }
      if (have) {
        fp.queue = ctx->d_queue[qi];
        fp.queue_shard_cap = shard_cap;
        fp.queue_plane = plane;
        fp.spill_max = t;
        const bool lists_now = ctx->scene.dm.enabled && ctx->scene.dm.lists;
        if (ctx->tail_occupancy_records != ctx->scene.image_records || ctx->tail_occupancy_blocks == 0 || ctx->tail_occupancy_lists != lists_now) {
          ctx->tail_occupancy_blocks = tail_blocks_per_cu(static_cast<size_t>(ctx->scene.image_records) * sizeof(float4), lists_now);
          ctx->tail_occupancy_records = ctx->scene.image_records;
          ctx->tail_occupancy_lists = lists_now;
        }
        const int per_cu_tail = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (ctx->tail_occupancy_blocks > 0 ? ctx->tail_occupancy_blocks : 4);
        lc.tail_blocks = per_cu_tail * ctx->cu_count;
        lc.tail_start = ctx->ev_tail[slot];
      }
    }
  }
  ctx->tail_ran[slot] = fp.spill_max != 0u;
  if (ctx->present_guard >= 0) {  // a presentation copy still reads the texels this launch (its resolve pass) overwrites
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->present_done[ctx->present_guard], 0));
    ctx->present_guard = -1;
  }
  // (the start event only serves the timing queries: single-frame launches in flight may go without, RT_OPT_TIMING = 0)
  ctx->timed[slot] = (ctx->timing != 0 || !pipelined || batch > 1u) ? 1 : 0;
  if (ctx->timed[slot]) RT_HIP(ctx, hipEventRecord(ctx->ev_start[slot], render_stream));
  int per_cu = ctx->blocks_per_cu;
  if (per_cu <= 0 && ctx->variant == 2) {  // only the resident-wave shape sizes its grid by occupancy
    if (ctx->occupancy_records != ctx->scene.image_records) {
      ctx->occupancy_blocks = persistent_blocks_per_cu(static_cast<size_t>(ctx->scene.image_records) * sizeof(float4));
      ctx->occupancy_records = ctx->scene.image_records;
    }
    per_cu = ctx->occupancy_blocks;
  }
  if (per_cu <= 0) per_cu = 4;
  lc.persistent_blocks = per_cu * ctx->cu_count;
  int variant_used = -1, stage_used = -1;
  const int lrc = launch_render(ctx->scene, fp, lc, render_stream, &variant_used, &stage_used);
  if (lrc != 0) return fail(ctx, RT_ERR_DEVICE, "render launch failed: %s", hipGetErrorString(static_cast<hipError_t>(lrc)));
  RT_HIP(ctx, hipEventRecord(ctx->ev_stop[slot], render_stream));
  if (pipelined)  // the main stream joins the renders in frame order: everything after this point sees frame k done
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_stop[slot], 0));
  ctx->rnd_used[rnd] = true;  // (rt_set_random records the copy's last use itself, when it is about to refill it)
  rnd_lock.unlock();
  if (staging != nullptr && launch_resolve(rp, ctx->stream, (ctx->flags & 16) != 0) != 0) return fail(ctx, RT_ERR_DEVICE, "resolve launch failed");
  RT_HIP(ctx, hipEventRecord(ctx->ev_resolved[slot], ctx->stream));
  if (pipelined) {  // the lane's sample buffer is free again once this launch's resolve has run
    ctx->resolve_slot[lane] = slot;
    ctx->resolve_pending[lane] = true;
    ctx->pipe_frames += 1;
  }
  if (ctx->comm) {
    const int grc = gather_frame(ctx, send);
    if (grc != RT_OK) return grc;
  }
  if (ctx->exchanges()) {  // (a member of an rt_multi: its owner gathers after all members have launched)
    ctx->last_send = send;
    ctx->comm_launches += 1;
  }
  {
    size_t queue_records = 0;
    for (size_t cap : ctx->queue_cap) queue_records += cap;
    ctx->stats.path_queue_bytes = static_cast<uint64_t>(queue_records) * sizeof(float4);
  }
  ctx->stats.last_variant = static_cast<uint32_t>(variant_used);
  ctx->stats.last_stage = static_cast<uint32_t>(stage_used);
  ctx->stats.grid_cells = ctx->scene.grid.enabled ? ctx->scene.grid.n_cells : 0u;
  ctx->stats.grid_items = ctx->scene.grid.enabled ? ctx->scene.grid.n_items : 0u;
  ctx->stats.dirmask_cells = ctx->scene.dm.enabled ? ctx->scene.dm.nx * ctx->scene.dm.ny * ctx->scene.dm.nz : 0u;
  ctx->stats.dirmask_bins = ctx->scene.dm.enabled ? ctx->scene.dm.bins : 0u;
  ctx->stats.dirmask_lists = (ctx->scene.dm.enabled && ctx->scene.dm.lists) ? ctx->dirlist_reach_used : 0u;
  if (fp.hits) {
    ctx->hits_valid = true;
    ctx->hits_rendered_limit = fp.reflection_limit;
  }
  ctx->timing_valid = true;
  ctx->stats.frames += 1;
  ctx->stats.last_batch = batch;
  ctx->stats.last_launches_in_flight = pipelined ? static_cast<uint32_t>(lanes) : 1u;
  return RT_OK;
}

int rt_render(rt_context *ctx, float jitter_x, float jitter_y, uint32_t progressive_count) {
  const float jitter[2] = {jitter_x, jitter_y};
  return render_frames(ctx, 1u, jitter, &progressive_count);
}

int rt_render_frames(rt_context *ctx, uint32_t n_frames, const void *framedata16) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!framedata16) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frames: NULL FrameData");
  if (n_frames < 1u || n_frames > kMaxBatch)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frames: %u frames, 1..%u allowed per call", n_frames, kMaxBatch);
  float jitter[2 * kMaxBatch];
  uint32_t counts[kMaxBatch];
  for (uint32_t f = 0; f < n_frames; ++f) {
    rt_wire_frame w;
    std::memcpy(&w, static_cast<const unsigned char *>(framedata16) + sizeof w * f, sizeof w);
    jitter[2 * f] = w.jitter[0];
    jitter[2 * f + 1] = w.jitter[1];
    counts[f] = w.progressive_count;
  }
  return render_frames(ctx, n_frames, jitter, counts);
}

int rt_render_frame(rt_context *ctx, const void *framedata16) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!framedata16) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_frame: NULL FrameData");
  rt_wire_frame f;
  std::memcpy(&f, framedata16, sizeof f);
  return rt_render(ctx, f.jitter[0], f.jitter[1], f.progressive_count);
}

int rt_sync(rt_context *ctx) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));  // the gather of the last launch too
  return RT_OK;
}

static const uint32_t *current_texels(rt_context *ctx) {  // caller-bound buffers take precedence
  if (!ctx) return nullptr;
  if (ctx->ext_rgba8) return ctx->ext_rgba8;
  return (ctx->exchanges() && ctx->last_send) ? ctx->d_rgba8_alt : ctx->d_rgba8;  // multi-GPU frame: the target of the last launch
}
int rt_read_rgba8(rt_context *ctx, void *dst, size_t bytes) {
  return read_back(ctx, current_texels(ctx), dst, bytes, 4, "rt_read_rgba8");
}
int rt_read_rgbf32(rt_context *ctx, float *dst, size_t bytes) {
  return read_back(ctx, ctx ? (ctx->ext_rgba8 ? ctx->ext_rgbf32 : ctx->d_rgbf32) : nullptr, dst, bytes, 12, "rt_read_rgbf32");
}
int rt_read_rgba8_rows(rt_context *ctx, uint32_t first_row, uint32_t n_rows, void *dst, size_t bytes) {
  return read_back(ctx, current_texels(ctx), dst, bytes, 4, "rt_read_rgba8_rows", first_row, n_rows);
}
int rt_read_rgbf32_rows(rt_context *ctx, uint32_t first_row, uint32_t n_rows, float *dst, size_t bytes) {
  return read_back(ctx, ctx ? (ctx->ext_rgba8 ? ctx->ext_rgbf32 : ctx->d_rgbf32) : nullptr, dst, bytes, 12,
                   "rt_read_rgbf32_rows", first_row, n_rows);
}
int rt_read_segments(rt_context *ctx, uint32_t *dst, size_t bytes) {
  return read_back(ctx, ctx ? ctx->d_segments : nullptr, dst, bytes, 4, "rt_read_segments");
}
int rt_read_hits(rt_context *ctx, int32_t *dst, size_t bytes) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  // The buffer holds the last RENDERED frame: its stride is that frame's reflection_limit, which a later
  // rt_set_scene may have changed (and the allocation only grows at the next render).
  if (ctx->d_hits && !ctx->hits_valid)
    return fail(ctx, RT_ERR_NOT_READY, "rt_read_hits: no frame rendered since the target was (re)allocated");
  if (ctx->d_hits && ctx->hits_rendered_limit != ctx->config.reflection_limit)
    return fail(ctx, RT_ERR_NOT_READY, "rt_read_hits: the last frame was rendered with reflection_limit %u, the scene now "
                                       "has %u; render a frame first", ctx->hits_rendered_limit, ctx->config.reflection_limit);
  return read_back(ctx, ctx->d_hits, dst, bytes, 4 * static_cast<size_t>(ctx->hits_rendered_limit), "rt_read_hits");
}

// Asynchronous presentation.  `src` (device memory of `bytes` bytes) is complete once everything queued on `producer` so far has
// run.  With snapshot = true the frame is first copied to a device buffer of the slot on the producer stream (a few
// microseconds) and goes to the pinned host buffer from there, on `copier`: the producer's next launches overwrite `src` in
// place (the target is the history) without waiting for the 0.16 ms the host copy of a 1080p frame takes, so consecutive
// frames' host copies run back to back at the link's rate.  snapshot = false (copier == producer): ordered by the stream itself.
static int present_from(rt_context *ctx, uint32_t slot, const void *src, size_t bytes, uint32_t rows, uint32_t width, hipStream_t producer,
                        hipStream_t copier, bool snapshot) {
  if (!ctx->present_done[slot]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->present_done[slot], hipEventDisableTiming));
  if (ctx->present_cap[slot] < bytes) {
    if (ctx->present_queued[slot]) RT_HIP(ctx, hipEventSynchronize(ctx->present_done[slot]));
    if (ctx->present_host[slot]) (void)hipHostFree(ctx->present_host[slot]);
    ctx->present_host[slot] = nullptr;
    ctx->present_cap[slot] = 0;
    RT_HIP(ctx, hipHostMalloc(&ctx->present_host[slot], bytes, hipHostMallocDefault));
    ctx->present_cap[slot] = bytes;
  }
  // the slot's previous copy (the caller has moved on to this frame: it no longer wants the old one) must not be overtaken
  if (ctx->present_queued[slot]) RT_HIP(ctx, hipStreamWaitEvent(producer, ctx->present_done[slot], 0));
  const void *from = src;
  if (snapshot) {
    if (ctx->present_dev_cap[slot] < bytes) {
      if (ctx->present_queued[slot]) RT_HIP(ctx, hipEventSynchronize(ctx->present_done[slot]));
      (void)hipFree(ctx->d_present[slot]);
      ctx->d_present[slot] = nullptr;
      ctx->present_dev_cap[slot] = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_present[slot], bytes));
      ctx->present_dev_cap[slot] = bytes;
    }
    if (launch_copy_texels(src, ctx->d_present[slot], bytes, producer) != 0) return fail(ctx, RT_ERR_DEVICE, "rt_present_async: the snapshot launch failed");
    from = ctx->d_present[slot];
  }
  if (copier != producer) {
    if (!ctx->present_ready) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->present_ready, hipEventDisableTiming));
    RT_HIP(ctx, hipEventRecord(ctx->present_ready, producer));
    RT_HIP(ctx, hipStreamWaitEvent(copier, ctx->present_ready, 0));
  }
  // (the DMA engine: a kernel storing into the pinned buffer instead moves 29 GB/s beside the rendering where this moves 41)
  RT_HIP(ctx, hipMemcpyAsync(ctx->present_host[slot], from, bytes, hipMemcpyDeviceToHost, copier));
  RT_HIP(ctx, hipEventRecord(ctx->present_done[slot], copier));
  ctx->present_bytes[slot] = bytes;
  ctx->present_rows[slot] = rows;
  ctx->present_width[slot] = width;
  ctx->present_queued[slot] = true;
  return RT_OK;
}

int rt_present_async(rt_context *ctx, uint32_t slot) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  const Range range("rt355 rt_present_async");
  if (slot >= rt_context::kPresentSlots) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_present_async: slot %u of %u", slot, rt_context::kPresentSlots);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->comm) {  // a communicator's frame is assembled on rank 0, on the communication stream: the copy queues up behind the gather
    if (ctx->comm_rank != 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_present_async: the frame is assembled on rank 0 only");
    if (!ctx->d_frame) return fail(ctx, RT_ERR_NOT_READY, "rt_present_async: no target allocated");
    return present_from(ctx, slot, ctx->d_frame, static_cast<size_t>(ctx->width) * ctx->height * 4, ctx->height, ctx->width, ctx->comm_stream,
                        ctx->comm_stream, false);
  }
  const uint32_t *src = current_texels(ctx);
  if (!src || !ctx->local_rows) return fail(ctx, RT_ERR_NOT_READY, "rt_present_async: no target allocated");
  if (!ctx->present_stream) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->present_stream, hipStreamNonBlocking));
  return present_from(ctx, slot, src, static_cast<size_t>(ctx->local_rows) * ctx->width * 4, ctx->local_rows, ctx->width, ctx->stream,
                      ctx->present_stream, true);
}

int rt_present_wait(rt_context *ctx, uint32_t slot, const void **texels, size_t *bytes, uint32_t *rows, uint32_t *width) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (slot >= rt_context::kPresentSlots || !texels) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_present_wait: slot %u / NULL", slot);
  if (!ctx->present_queued[slot]) return fail(ctx, RT_ERR_NOT_READY, "rt_present_wait: nothing was queued into slot %u (rt_present_async)", slot);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipEventSynchronize(ctx->present_done[slot]));
  ctx->present_queued[slot] = false;
  *texels = ctx->present_host[slot];
  if (bytes) *bytes = ctx->present_bytes[slot];
  if (rows) *rows = ctx->present_rows[slot];
  if (width) *width = ctx->present_width[slot];
  return RT_OK;
}

int rt_keep_batch_frames(rt_context *ctx, uint32_t every) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  ctx->keep_every = every;
  return RT_OK;
}

int rt_read_batch_frame(rt_context *ctx, uint32_t frame_in_launch, void *dst, size_t bytes) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!dst) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_batch_frame: dst is NULL");
  if (!ctx->kept_frames || !ctx->d_kept)
    return fail(ctx, RT_ERR_NOT_READY, "rt_read_batch_frame: the last launch kept no frames (rt_keep_batch_frames, rt_render_frames)");
  if ((frame_in_launch + 1u) % ctx->kept_every_last != 0u || frame_in_launch / ctx->kept_every_last >= ctx->kept_frames)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_batch_frame: frame %u was not kept (every %u, %u kept)", frame_in_launch,
                ctx->kept_every_last, ctx->kept_frames);
  const size_t px = static_cast<size_t>(ctx->local_rows) * ctx->width;
  if (bytes != px * 4) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_read_batch_frame: %zu bytes given, %zu expected", bytes, px * 4);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_kept + static_cast<size_t>(frame_in_launch / ctx->kept_every_last) * px, bytes,
                             hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_write_history_rgba8(rt_context *ctx, const void *src, size_t bytes) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (!src) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_write_history_rgba8: NULL");
  if (!ctx->d_rgba8) return fail(ctx, RT_ERR_NOT_READY, "rt_write_history_rgba8: no target");
  const size_t need = static_cast<size_t>(ctx->local_rows) * ctx->width * 4;
  if (bytes != need) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_write_history_rgba8: %zu bytes given, %zu expected", bytes, need);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t *history = ctx->ext_rgba8 ? (ctx->ext_hist_rgba8 ? ctx->ext_hist_rgba8 : ctx->ext_rgba8) : ctx->d_rgba8;
  if (ctx->exchanges() && !ctx->ext_rgba8) {  // the next launch reads the buffer it does not write ...
    const int idx = (ctx->comm_launches & 1u) ? 0 : 1;
    history = idx ? ctx->d_rgba8_alt : ctx->d_rgba8;
    // ... which is the one the previous launch wrote and whose gather may still be sending from it
    if (ctx->gather_pending[idx]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gather_done[idx], 0));
  }
  RT_HIP(ctx, hipMemcpyAsync(history, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_get_device_buffers(rt_context *ctx, rt_device_buffers *out) {
  if (!ctx || !out) return RT_ERR_INVALID_ARGUMENT;
  out->rgba8 = const_cast<uint32_t *>(current_texels(ctx));  // comm mode: the target of the last launch
  out->rgbf32 = ctx->ext_rgba8 ? ctx->ext_rgbf32 : ctx->d_rgbf32;
  out->segments = ctx->d_segments;
  out->hits = ctx->d_hits;
  out->stream = ctx->stream;
  out->width = ctx->width;
  out->local_rows = ctx->local_rows;
  return RT_OK;
}

int rt_get_stats(rt_context *ctx, rt_stats *out) {
  if (!ctx || !out) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->timing_valid) {
    const uint32_t slot = static_cast<uint32_t>((ctx->stats.frames - 1) % rt_context::kRing);
    std::vector<unsigned long long> line(rt_context::kCounterWords);
    RT_HIP(ctx, hipMemcpy(line.data(), ctx->d_counters + rt_context::kCounterWords * static_cast<size_t>(slot),
                          line.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long c[4] = {0, 0, 0, 0};
    for (uint32_t sh = 0; sh < kTotalShards; ++sh)
      for (int w = 0; w < 4; ++w) c[w] += line[sh * kTotalShardStride + w];
    ctx->stats.last_cull_mismatches = c[3];
    ctx->stats.last_segments = c[0];
    ctx->stats.last_rough_hits = c[1];
    ctx->stats.last_wave_iterations = c[2];
    float ms = 0.0f;
    if (ctx->timed[slot]) RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
    ctx->stats.last_kernel_ms = ms;
    RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_stop[slot], ctx->ev_resolved[slot]));
    ctx->stats.last_resolve_ms = ms;
  }
  ctx->stats.last_pixels = static_cast<uint64_t>(ctx->local_rows) * ctx->width * (ctx->stats.last_batch ? ctx->stats.last_batch : 1u);
  ctx->stats.width = ctx->width;
  ctx->stats.height = ctx->height;
  ctx->stats.local_rows = ctx->local_rows;
  *out = ctx->stats;
  return RT_OK;
}

int rt_get_counter_words(rt_context *ctx, uint64_t *out16) {
  if (!ctx || !out16) return RT_ERR_INVALID_ARGUMENT;
  if (!ctx->timing_valid) return fail(ctx, RT_ERR_NOT_READY, "rt_get_counter_words: nothing rendered yet");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t slot = static_cast<uint32_t>((ctx->stats.frames - 1) % rt_context::kRing);
  std::vector<unsigned long long> line(rt_context::kCounterWords);
  RT_HIP(ctx, hipMemcpy(line.data(), ctx->d_counters + rt_context::kCounterWords * static_cast<size_t>(slot),
                        line.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (uint32_t w = 0; w < kTotalShardStride; ++w) {
    out16[w] = 0;
    for (uint32_t sh = 0; sh < kTotalShards; ++sh) out16[w] += line[sh * kTotalShardStride + w];
  }
  unsigned long long events[8];  // event-count build: counts since the previous call — words 12..15, and the direction-mask
                                 // counts in words 4..7 (the phase sums, which that build does not fill)
  if (prof_counts(events, true) == 0) {
    for (int k = 0; k < 4; ++k) out16[12 + k] = events[k];
    if (events[4] != 0)
      for (int k = 0; k < 4; ++k) out16[4 + k] = events[4 + k];
  }
  return RT_OK;
}

int rt_get_frame_history(rt_context *ctx, uint32_t n, uint64_t *segments, uint64_t *rough_hits, float *kernel_ms) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  if (n > rt_context::kRing / 2 || n > ctx->stats.frames)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_get_frame_history: %u frames asked, %llu rendered, ring keeps %u", n,
                static_cast<unsigned long long>(ctx->stats.frames), rt_context::kRing / 2);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<unsigned long long> ring(rt_context::kRing * rt_context::kCounterWords);
  RT_HIP(ctx, hipMemcpy(ring.data(), ctx->d_counters, ring.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < n; ++k) {  // oldest first
    const uint32_t slot = static_cast<uint32_t>((ctx->stats.frames - n + k) % rt_context::kRing);
    unsigned long long c[2] = {0, 0};
    for (uint32_t sh = 0; sh < kTotalShards; ++sh) {
      c[0] += ring[rt_context::kCounterWords * static_cast<size_t>(slot) + sh * kTotalShardStride];
      c[1] += ring[rt_context::kCounterWords * static_cast<size_t>(slot) + sh * kTotalShardStride + 1];
    }
    if (segments) segments[k] = c[0];
    if (rough_hits) rough_hits[k] = c[1];
    if (kernel_ms) {
      kernel_ms[k] = 0.0f;  // (a launch without timing events, RT_OPT_TIMING = 0)
      if (ctx->timed[slot]) RT_HIP(ctx, hipEventElapsedTime(&kernel_ms[k], ctx->ev_start[slot], ctx->ev_stop[slot]));
    }
  }
  return RT_OK;
}

int rt_get_resolve_history(rt_context *ctx, uint32_t n, float *resolve_ms) {
  if (!ctx || !resolve_ms) return RT_ERR_INVALID_ARGUMENT;
  if (n > rt_context::kRing / 2 || n > ctx->stats.frames)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_get_resolve_history: %u launches asked, %llu made, ring keeps %u", n,
                static_cast<unsigned long long>(ctx->stats.frames), rt_context::kRing / 2);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t k = 0; k < n; ++k) {  // oldest first
    const uint32_t slot = static_cast<uint32_t>((ctx->stats.frames - n + k) % rt_context::kRing);
    RT_HIP(ctx, hipEventElapsedTime(&resolve_ms[k], ctx->ev_stop[slot], ctx->ev_resolved[slot]));
  }
  return RT_OK;
}

int rt_get_tail_history(rt_context *ctx, uint32_t n, float *tail_ms) {
  if (!ctx || !tail_ms) return RT_ERR_INVALID_ARGUMENT;
  if (n > rt_context::kRing / 2 || n > ctx->stats.frames)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_get_tail_history: %u launches asked, %llu made, ring keeps %u", n,
                static_cast<unsigned long long>(ctx->stats.frames), rt_context::kRing / 2);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t k = 0; k < n; ++k) {  // oldest first
    const uint32_t slot = static_cast<uint32_t>((ctx->stats.frames - n + k) % rt_context::kRing);
    tail_ms[k] = 0.0f;
    if (ctx->tail_ran[slot]) RT_HIP(ctx, hipEventElapsedTime(&tail_ms[k], ctx->ev_tail[slot], ctx->ev_stop[slot]));
  }
  return RT_OK;
}

int rt_get_launch_span(rt_context *ctx, uint32_t n, float *span_ms) {
  if (!ctx || !span_ms || n == 0) return RT_ERR_INVALID_ARGUMENT;
  if (n > rt_context::kRing / 2 || n > ctx->stats.frames)
    return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_get_launch_span: %u launches asked, %llu made, ring keeps %u", n,
                static_cast<unsigned long long>(ctx->stats.frames), rt_context::kRing / 2);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t first = static_cast<uint32_t>((ctx->stats.frames - n) % rt_context::kRing);
  const uint32_t last = static_cast<uint32_t>((ctx->stats.frames - 1) % rt_context::kRing);
  *span_ms = 0.0f;
  if (ctx->timed[first]) RT_HIP(ctx, hipEventElapsedTime(span_ms, ctx->ev_start[first], ctx->ev_stop[last]));
  return RT_OK;
}

int rt_set_stream(rt_context *ctx, void *hip_stream) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  ctx->owns_stream = false;
  return RT_OK;
}

int rt_bind_outputs(rt_context *ctx, void *d_rgba8, void *d_history_rgba8, void *d_rgbf32) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  ctx->ext_rgba8 = static_cast<uint32_t *>(d_rgba8);
  ctx->ext_hist_rgba8 = static_cast<uint32_t *>(d_history_rgba8);
  ctx->ext_rgbf32 = static_cast<float *>(d_rgbf32);
  return RT_OK;
}

const char *rt_last_error(rt_context *ctx) { return ctx ? ctx->error.c_str() : "rt_context is NULL"; }

int rt_selftest_math_unary(rt_context *ctx, int op, const float *x, float *y, size_t n) {
  if (!ctx || !x || !y) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  float *dx = nullptr, *dy = nullptr;
  hipError_t e = hipMalloc(&dx, n * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&dy, n * 4 + 4);
  if (e == hipSuccess) e = hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
  int rc = 0;
  if (e == hipSuccess) {
    rc = launch_math_unary(op, dx, dy, n, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (rc == 0 && e == hipSuccess) e = hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dx);
  (void)hipFree(dy);
  if (rc != 0 || e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "math self-test failed");
  return RT_OK;
}

int rt_selftest_math_binary(rt_context *ctx, int op, const float *a, const float *b, float *y, size_t n) {
  if (!ctx || !a || !b || !y) return RT_ERR_INVALID_ARGUMENT;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  float *da = nullptr, *db = nullptr, *dy = nullptr;
  hipError_t e = hipMalloc(&da, n * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&db, n * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&dy, n * 4 + 4);
  if (e == hipSuccess) e = hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
  int rc = 0;
  if (e == hipSuccess) {
    rc = launch_math_binary(op, da, db, dy, n, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (rc == 0 && e == hipSuccess) e = hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dy);
  if (rc != 0 || e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "math self-test failed");
  return RT_OK;
}

// ---- the multi-GPU frame driven by ONE process (rt_multi_*) -------------------------------------------------------------
// The reference is one process with one loop (app.rs:208-254).  An rt_multi owns one context per device, gives context i
// the row strips s with s % n == i, forwards every setter to all of them, launches every render on all of them (the
// launches are asynchronous: the devices work side by side) and gathers on device 0: peer copies of the members' strips
// (hipMemcpyPeerAsync: one xGMI link per peer) into the layout rank 0 of a communicator receives, then the same assembly
// kernel.  Targets alternate exactly as with a communicator, so the gather of launch k runs beside launch k + 1.
struct rt_multi {
  std::vector<rt_context *> ctx;
  uint32_t strip_rows = 8;
  int device0 = 0;
  hipStream_t gather_stream = nullptr;
  hipEvent_t done[2] = {nullptr, nullptr};  // the gather that read target b has finished (recorded on gather_stream)
  // rt_multi_set_random / rt_multi_generate_random may come from another thread (the reference refreshes the texture every
  // 4 ms on a thread of its own, random_texture.rs:56-61): the texture changes on ALL devices between two launches, never
  // while one launch is being issued device by device — a frame sees one texture in every strip
  std::mutex random_mutex;
  // One host thread per device issues that device's launches: a launch costs ~20 us of host time, which one thread pays
  // n times per frame in turn (8 devices: 0.2 ms, more than a device's share of the frame takes to render).
  std::vector<std::thread> workers;
  std::mutex job_mutex;
  std::condition_variable job_cv, done_cv;
  uint64_t job_generation = 0;
  uint32_t job_pending = 0;
  // the same two counters for the threads that SPIN for a few tens of microseconds before they sleep on the condition variables:
  // in a render loop the next frame's job arrives within microseconds of the last one's end, and a futex wake-up costs more than
  // the job's own HIP calls
  std::atomic<uint64_t> job_generation_hint{0};
  std::atomic<uint32_t> job_pending_hint{0};
  bool quit = false;
  std::function<int(rt_context *)> job;
  std::vector<int> job_result;
  uint32_t peers_direct = 0, peers_staged = 0;  // members on another device than the first: with / without direct peer access to it
  std::string error;
};

static int multi_fail(rt_multi *m, int code, const char *what, rt_context *from = nullptr) {
  if (m) m->error = std::string(what) + (from ? std::string(": ") + from->error : std::string());
  return code;
}

static void multi_worker(rt_multi *m, uint32_t index) {
  uint64_t seen = 0;
  for (;;) {
    std::function<int(rt_context *)> job;
    {
      for (int spin = 0; spin < 4000 && m->job_generation_hint.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
      std::unique_lock<std::mutex> lock(m->job_mutex);
      m->job_cv.wait(lock, [&] { return m->quit || m->job_generation != seen; });
      if (m->quit) return;
      seen = m->job_generation;
      job = m->job;
    }
    const int rc = job(m->ctx[index]);
    {
      std::lock_guard<std::mutex> lock(m->job_mutex);
      m->job_result[index] = rc;
      m->job_pending -= 1;
      m->job_pending_hint.store(m->job_pending, std::memory_order_release);
    }
    m->done_cv.notify_one();
  }
}

// `fn` on every member, each on its device's own host thread (inline when there is one device); the first failure wins
static int multi_run(rt_multi *m, const char *what, const std::function<int(rt_context *)> &fn) {
  if (m->workers.empty()) {
    for (rt_context *c : m->ctx) {
      const int rc = fn(c);
      if (rc != RT_OK) return multi_fail(m, rc, what, c);
    }
    return RT_OK;
  }
  {
    std::lock_guard<std::mutex> lock(m->job_mutex);
    m->job = fn;
    m->job_pending = static_cast<uint32_t>(m->ctx.size());
    m->job_pending_hint.store(m->job_pending, std::memory_order_release);
    m->job_generation += 1;
    m->job_generation_hint.store(m->job_generation, std::memory_order_release);
  }
  m->job_cv.notify_all();
  for (int spin = 0; spin < 20000 && m->job_pending_hint.load(std::memory_order_acquire) != 0u; ++spin) __builtin_ia32_pause();
  {
    std::unique_lock<std::mutex> lock(m->job_mutex);
    m->done_cv.wait(lock, [&] { return m->job_pending == 0; });
  }
  for (size_t i = 0; i < m->ctx.size(); ++i)
    if (m->job_result[i] != RT_OK) return multi_fail(m, m->job_result[i], what, m->ctx[i]);
  return RT_OK;
}

// what a member's own thread does after its launch: mark the frame's strips ready for the gather (one call on that thread
// instead of two per member on the caller's)
static int multi_mark_ready(rt_context *c) {
  if (!c->multi_member) return RT_OK;
  if (hipEventRecord(c->frame_ready[c->last_send], c->stream) != hipSuccess) return fail(c, RT_ERR_DEVICE, "rt_multi: recording a frame event failed");
  return RT_OK;
}

// after every member has launched: strips -> device 0 -> frame
static int multi_gather(rt_multi *m) {
  const uint32_t n = static_cast<uint32_t>(m->ctx.size());
  if (n == 1u) return RT_OK;  // one device: its target is the frame
  const Range range("rt355 gather (peer copies)");
  rt_context *root = m->ctx[0];
  const int b = root->last_send;
  for (uint32_t i = 0; i < n; ++i)
    if (m->ctx[i]->last_send != b) return multi_fail(m, RT_ERR_DEVICE, "rt_multi: the members' targets are out of step");
  if (hipSetDevice(m->device0) != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, "rt_multi: hipSetDevice failed");
  for (uint32_t i = 0; i < n; ++i)
    if (hipStreamWaitEvent(m->gather_stream, m->ctx[i]->frame_ready[b], 0) != hipSuccess)
      return multi_fail(m, RT_ERR_DEVICE, "rt_multi: waiting for a frame event failed");
  AssembleParams ap{};
  fill_assemble_layout(ap, root->width, root->height, n, m->strip_rows);
  const size_t row_bytes = static_cast<size_t>(root->width) * 4;
  for (uint32_t i = 1; i < n; ++i) {
    rt_context *c = m->ctx[i];
    if (!c->local_rows) continue;
    const uint32_t *src = b ? c->d_rgba8_alt : c->d_rgba8;
    const bool dry = (root->flags & 16) != 0;  // RT_OPT_FLAGS bit 4: the calls of a frame with next to none of its work
    if (hipMemcpyPeerAsync(root->d_stack + ap.stack_offset[i], m->device0, src, c->device, (dry ? 1u : c->local_rows) * row_bytes, m->gather_stream) !=
        hipSuccess)
      return multi_fail(m, RT_ERR_DEVICE, "rt_multi: hipMemcpyPeerAsync failed");
  }
  ap.frame = root->d_frame;
  ap.own = b ? root->d_rgba8_alt : root->d_rgba8;
  ap.stack = root->d_stack;
  if (launch_assemble(ap, m->gather_stream, (root->flags & 16) != 0) != 0) return multi_fail(m, RT_ERR_DEVICE, "rt_multi: assemble launch failed");
  if (hipEventRecord(m->done[b], m->gather_stream) != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, "rt_multi: recording the gather event failed");
  for (uint32_t i = 0; i < n; ++i) m->ctx[i]->gather_pending[b] = true;  // target b is written again only after this gather
  return RT_OK;
}

#define RT_MULTI_EACH(m, what, call)                                              \
  do {                                                                            \
    if (!(m)) return RT_ERR_INVALID_ARGUMENT;                                     \
    for (rt_context * c_ : (m)->ctx) {                                            \
      const int rc_ = (call);                                                     \
      if (rc_ != RT_OK) return multi_fail((m), rc_, what, c_);                    \
    }                                                                             \
  } while (0)

int rt_multi_destroy(rt_multi *m) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(m->job_mutex);
    m->quit = true;
  }
  m->job_cv.notify_all();
  for (std::thread &t : m->workers) t.join();
  for (rt_context *c : m->ctx) {  // everything in flight first: the members' streams, then the gathers that read them
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
  }
  (void)hipSetDevice(m->device0);
  if (m->gather_stream) (void)hipStreamSynchronize(m->gather_stream);
  for (rt_context *c : m->ctx) {
    c->gather_done[0] = c->gather_done[1] = nullptr;  // the rt_multi's events, not the context's
    c->gather_pending[0] = c->gather_pending[1] = false;
    (void)rt_destroy(c);
  }
  (void)hipSetDevice(m->device0);
  for (int b = 0; b < 2; ++b)
    if (m->done[b]) (void)hipEventDestroy(m->done[b]);
  if (m->gather_stream) (void)hipStreamDestroy(m->gather_stream);
  delete m;
  return RT_OK;
}

int rt_multi_create(const int *device_ids, uint32_t n_devices, uint32_t strip_rows, rt_multi **out) {
  if (!out) return RT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!device_ids || !n_devices || n_devices > kMaxRanks || !strip_rows) return RT_ERR_INVALID_ARGUMENT;
  rt_multi *m = new (std::nothrow) rt_multi();
  if (!m) return RT_ERR_OUT_OF_MEMORY;
  m->strip_rows = strip_rows;
  m->device0 = device_ids[0];
  auto bail = [&](int code) {
    rt_multi_destroy(m);
    return code;
  };
  for (uint32_t i = 0; i < n_devices; ++i) {
    rt_context *c = nullptr;
    const int rc = rt_create(device_ids[i], &c);
    if (rc != RT_OK) return bail(rc);
    m->ctx.push_back(c);
    c->out_flags = RT_OUT_RGBA8;
    if (n_devices == 1u) continue;  // one device: a plain context, no strips, no gather
    c->multi_member = true;
    c->rank = c->comm_rank = i;
    c->world = c->comm_world = n_devices;
    c->strip_rows = strip_rows;
    for (int b = 0; b < 2; ++b)
      if (hipEventCreateWithFlags(&c->frame_ready[b], hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_DEVICE);
  }
  if (hipSetDevice(m->device0) != hipSuccess) return bail(RT_ERR_DEVICE);
  if (hipStreamCreateWithFlags(&m->gather_stream, hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_DEVICE);
  for (int b = 0; b < 2; ++b)
    if (hipEventCreateWithFlags(&m->done[b], hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_DEVICE);
  for (rt_context *c : m->ctx) {
    if (!c->multi_member) continue;
    c->gather_done[0] = m->done[0];
    c->gather_done[1] = m->done[1];
    if (c->device != m->device0) {  // direct peer copies where the devices can (xGMI); otherwise the runtime stages them through the host
      int can = 0;
      bool direct = false;
      if (hipDeviceCanAccessPeer(&can, m->device0, c->device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(c->device, 0);
        direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
      }
      (void)hipGetLastError();
      (direct ? m->peers_direct : m->peers_staged) += 1u;  // reported by rt_multi_peer_status: a staged gather is visible, not silent
    }
  }
  if (n_devices > 1u) {
    m->job_result.assign(n_devices, RT_OK);
    try {
      for (uint32_t i = 0; i < n_devices; ++i) m->workers.emplace_back(multi_worker, m, i);
    } catch (...) {
      return bail(RT_ERR_OUT_OF_MEMORY);
    }
  }
  *out = m;
  return RT_OK;
}

int rt_multi_peer_status(rt_multi *m, uint32_t *direct, uint32_t *staged) {
  if (!m || !direct || !staged) return RT_ERR_INVALID_ARGUMENT;
  *direct = m->peers_direct;
  *staged = m->peers_staged;
  return RT_OK;
}
uint32_t rt_multi_device_count(rt_multi *m) { return m ? static_cast<uint32_t>(m->ctx.size()) : 0u; }
rt_context *rt_multi_context(rt_multi *m, uint32_t index) { return (m && index < m->ctx.size()) ? m->ctx[index] : nullptr; }
const char *rt_multi_last_error(rt_multi *m) { return m ? m->error.c_str() : "rt_multi is NULL"; }

int rt_multi_set_skybox(rt_multi *m, uint32_t width, uint32_t height, const float *rgba32f) {
  RT_MULTI_EACH(m, "rt_multi_set_skybox", rt_set_skybox(c_, width, height, rgba32f));
  return RT_OK;
}
int rt_multi_set_skybox_exr(rt_multi *m, const char *path) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  uint32_t w = 0, h = 0;
  float *texels = nullptr;
  const int rc = rt_exr_read_rgba32f(path, &w, &h, &texels);  // decoded once, uploaded to every device
  if (rc != RT_OK) return multi_fail(m, rc, rt_exr_last_error());
  const int rc2 = rt_multi_set_skybox(m, w, h, texels);
  rt_exr_free(texels);
  return rc2;
}
int rt_multi_set_random(rt_multi *m, uint32_t width, uint32_t height, const float *r32f) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(m->random_mutex);
  RT_MULTI_EACH(m, "rt_multi_set_random", rt_set_random(c_, width, height, r32f));
  return RT_OK;
}
int rt_multi_generate_random(rt_multi *m, uint32_t width, uint32_t height, uint64_t seed) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(m->random_mutex);
  RT_MULTI_EACH(m, "rt_multi_generate_random", rt_generate_random(c_, width, height, seed));
  return RT_OK;
}
int rt_multi_set_scene(rt_multi *m, const void *objects, uint32_t object_count, const void *lights, uint32_t light_count,
                       const void *config112) {
  RT_MULTI_EACH(m, "rt_multi_set_scene", rt_set_scene(c_, objects, object_count, lights, light_count, config112));
  return RT_OK;
}
int rt_multi_resize(rt_multi *m, uint32_t width, uint32_t height) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(m->device0);
  (void)hipStreamSynchronize(m->gather_stream);  // a gather may still read the buffers about to be replaced
  RT_MULTI_EACH(m, "rt_multi_resize", rt_resize(c_, width, height));
  return RT_OK;
}
int rt_multi_set_accum_mode(rt_multi *m, int mode) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(m->device0);
  (void)hipStreamSynchronize(m->gather_stream);
  RT_MULTI_EACH(m, "rt_multi_set_accum_mode", rt_set_accum_mode(c_, mode));
  return RT_OK;
}
int rt_multi_set_option(rt_multi *m, int option, int value) {
  RT_MULTI_EACH(m, "rt_multi_set_option", rt_set_option(c_, option, value));
  return RT_OK;
}
int rt_multi_render_frames(rt_multi *m, uint32_t n_frames, const void *framedata16) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(m->random_mutex);  // every device launches with the same random texture
    const int rc = multi_run(m, "rt_multi_render_frames", [=](rt_context *c) {
      const int r = rt_render_frames(c, n_frames, framedata16);
      return r == RT_OK ? multi_mark_ready(c) : r;
    });
    if (rc != RT_OK) return rc;
  }
  return multi_gather(m);
}
int rt_multi_render_frame(rt_multi *m, const void *framedata16) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(m->random_mutex);
    const int rc = multi_run(m, "rt_multi_render_frame", [=](rt_context *c) {
      const int r = rt_render_frame(c, framedata16);
      return r == RT_OK ? multi_mark_ready(c) : r;
    });
    if (rc != RT_OK) return rc;
  }
  return multi_gather(m);
}
int rt_multi_render(rt_multi *m, float jitter_x, float jitter_y, uint32_t progressive_count) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(m->random_mutex);
    const int rc = multi_run(m, "rt_multi_render", [=](rt_context *c) {
      const int r = rt_render(c, jitter_x, jitter_y, progressive_count);
      return r == RT_OK ? multi_mark_ready(c) : r;
    });
    if (rc != RT_OK) return rc;
  }
  return multi_gather(m);
}
int rt_multi_sync(rt_multi *m) {
  RT_MULTI_EACH(m, "rt_multi_sync", rt_sync(c_));
  if (hipSetDevice(m->device0) != hipSuccess || hipStreamSynchronize(m->gather_stream) != hipSuccess)
    return multi_fail(m, RT_ERR_DEVICE, "rt_multi_sync: the gather stream failed");
  return RT_OK;
}
int rt_multi_read_frame_rgba8(rt_multi *m, void *dst, size_t bytes) {
  if (!m || !dst) return RT_ERR_INVALID_ARGUMENT;
  rt_context *root = m->ctx[0];
  if (m->ctx.size() == 1u) {
    const int rc = rt_read_rgba8(root, dst, bytes);
    return rc == RT_OK ? RT_OK : multi_fail(m, rc, "rt_multi_read_frame_rgba8", root);
  }
  if (!root->d_frame) return multi_fail(m, RT_ERR_NOT_READY, "rt_multi_read_frame_rgba8: no target (rt_multi_resize)");
  const size_t need = static_cast<size_t>(root->width) * root->height * 4;
  if (bytes != need) return multi_fail(m, RT_ERR_INVALID_ARGUMENT, "rt_multi_read_frame_rgba8: wrong size");
  if (hipSetDevice(m->device0) != hipSuccess ||
      hipMemcpyAsync(dst, root->d_frame, bytes, hipMemcpyDeviceToHost, m->gather_stream) != hipSuccess ||  // behind the last gather
      hipStreamSynchronize(m->gather_stream) != hipSuccess)
    return multi_fail(m, RT_ERR_DEVICE, "rt_multi_read_frame_rgba8: copy failed");
  return RT_OK;
}
int rt_multi_present_async(rt_multi *m, uint32_t slot) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  rt_context *root = m->ctx[0];
  if (m->ctx.size() == 1u) {
    const int rc = rt_present_async(root, slot);
    return rc == RT_OK ? RT_OK : multi_fail(m, rc, "rt_multi_present_async", root);
  }
  if (slot >= rt_context::kPresentSlots) return multi_fail(m, RT_ERR_INVALID_ARGUMENT, "rt_multi_present_async: no such slot");
  if (!root->d_frame) return multi_fail(m, RT_ERR_NOT_READY, "rt_multi_present_async: no target (rt_multi_resize)");
  if (hipSetDevice(m->device0) != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, "rt_multi_present_async: hipSetDevice failed");
  // behind the last gather on the gather stream: the next gather's assembly queues up behind the copy in turn
  const int rc = present_from(root, slot, root->d_frame, static_cast<size_t>(root->width) * root->height * 4, root->height, root->width,
                              m->gather_stream, m->gather_stream, false);
  return rc == RT_OK ? RT_OK : multi_fail(m, rc, "rt_multi_present_async", root);
}
int rt_multi_present_wait(rt_multi *m, uint32_t slot, const void **texels, size_t *bytes, uint32_t *rows, uint32_t *width) {
  if (!m) return RT_ERR_INVALID_ARGUMENT;
  rt_context *root = m->ctx[0];
  if (hipSetDevice(m->device0) != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, "rt_multi_present_wait: hipSetDevice failed");
  const int rc = rt_present_wait(root, slot, texels, bytes, rows, width);
  return rc == RT_OK ? RT_OK : multi_fail(m, rc, "rt_multi_present_wait", root);
}
int rt_multi_get_stats(rt_multi *m, rt_stats *out) {
  if (!m || !out) return RT_ERR_INVALID_ARGUMENT;
  rt_stats sum{};
  bool first = true;
  for (rt_context *c : m->ctx) {
    rt_stats st{};
    const int rc = rt_get_stats(c, &st);
    if (rc != RT_OK) return multi_fail(m, rc, "rt_multi_get_stats", c);
    if (first) {
      sum = st;
      first = false;
      continue;
    }
    sum.last_segments += st.last_segments;
    sum.last_rough_hits += st.last_rough_hits;
    sum.last_pixels += st.last_pixels;
    sum.local_rows += st.local_rows;
    sum.last_wave_iterations += st.last_wave_iterations;
    sum.last_cull_mismatches += st.last_cull_mismatches;
    if (st.last_kernel_ms > sum.last_kernel_ms) sum.last_kernel_ms = st.last_kernel_ms;  // the devices run side by side
  }
  *out = sum;
  return RT_OK;
}

}  // extern "C"

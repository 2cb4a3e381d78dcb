/* rt355.h — C-ABI of the MI355X-native per-pixel ray-tracing path.
 *
 * This library replaces what the reference's `src/gpu` module does for
 * `App::render_scene` (src/app.rs:208-254): the wgpu bind group + full-screen
 * draw whose fragment shader is the ray tracer (src/shaders/frag.hlsl:48-66)
 * become one hand-written HIP kernel for gfx950.  Each entry point below names
 * the reference call it stands in for.  The wire formats of the blobs are in
 * rt355_layout.h.
 *
 * Conventions: every function returns 0 on success or a
 * negative rt_status; no C++ exception crosses the boundary; every pointer is
 * borrowed for the duration of the call only (inputs are copied before the
 * call returns); a context is driven from one render thread, except
 * rt_set_random which may be called concurrently from another thread (the
 * reference's RandomTexture thread, src/gpu/random_texture.rs:56-61).
 * There is no CPU fallback: rt_create fails with RT_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#include "rt355_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT355_ABI_VERSION 5
#define RT355_API __attribute__((visibility("default")))

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = -1, /* NULL / zero size / size mismatch                      */
  RT_ERR_NO_DEVICE = -2,        /* no HIP device, or device id out of range              */
  RT_ERR_NOT_READY = -3,        /* render before skybox / random / scene / size were set */
  RT_ERR_DEVICE = -4,           /* a HIP call failed; text in rt_last_error              */
  RT_ERR_OUT_OF_MEMORY = -5
} rt_status;

typedef struct rt_context rt_context;

/* Accumulation modes (App. B D7). */
enum {
  RT_ACCUM_SRGB8 = 0, /* reference-exact: history is the stored Rgba8UnormSrgb texel */
  RT_ACCUM_FLOAT32 = 1 /* linear float history (quality mode, not in the reference)  */
};

/* Optional per-pixel outputs, OR-ed into rt_set_outputs. */
enum {
  RT_OUT_RGBA8 = 1,    /* always produced (it is also the sRGB8 history)                    */
  RT_OUT_RGBF32 = 2,   /* linear float value before the sRGB store (parity / HDR consumers) */
  RT_OUT_SEGMENTS = 4, /* executed iterations of frag.hlsl:36-43 per pixel                  */
  RT_OUT_HITS = 8      /* object_index per iteration (reflection_limit ints per pixel)      */
};

typedef struct rt_stats {
  uint64_t frames;            /* rt_render calls completed                                  */
  uint64_t last_segments;     /* ray segments (bounce-loop iterations) of the last launch   */
  uint64_t last_rough_hits;   /* segments of the last frame that drew random() (rough hit)  */
  uint64_t last_pixels;       /* pixels this context rendered in the last frame             */
  float last_kernel_ms;       /* HIP-event time of the last frame's render kernel           */
  uint32_t width, height;     /* full frame                                                 */
  uint32_t local_rows;        /* rows owned by this context under the current partition     */
  uint32_t object_count, sphere_count, plane_count;
  uint32_t last_batch;        /* frames the last launch rendered (rt_render_frames; 1 for rt_render)  */
  uint64_t last_wave_iterations; /* variant 2: object scans executed by all waves (64 lanes each);
                                    last_segments / (64 * this) = lane occupancy of the scan */
  uint64_t last_cull_mismatches; /* RT_OPT_CULL = 2: primary rays whose culled scan disagreed with the
                                    full scan in the last frame (must be 0)                    */
  float last_resolve_ms;      /* HIP-event time of the last launch's history-resolve pass (n_frames > 1) */
  uint32_t last_variant;      /* launch shape the last launch used (RT_OPT_VARIANT values 0..4)       */
  uint32_t last_stage;        /* render_tiles (variant 0/1): what that launch staged into LDS: 0 nothing, 1 sphere
                                 records, 2 the whole scene image, 3 sphere records + the sphere grid (secondary rays
                                 walk the grid); render_blocks (variant 3): 10 * waves per workgroup + waves per SIMD
                                 its registers are budgeted for (27 multi-frame launches, 25 single frames), + 100 when tail_stream followed
                                 it (RT_OPT_TAIL_SPILL); 40 for variant 4 */
  uint32_t grid_cells, grid_items; /* the current scene's sphere grid (0 = none: small scene, or not buildable) */
  uint32_t dirmask_cells, dirmask_bins; /* the current scene's direction-mask table (RT_OPT_DIRMASK) or candidate-list table
                                           (RT_OPT_DIRLIST): cells and direction bins per cell; 0 = none              */
  uint32_t tail_spill_unavailable; /* 1: a launch wanted to spill its tail (RT_OPT_TAIL_SPILL) but the path queue could not be allocated —
                                      it rendered without the spill (same frame, slower); 0 otherwise                          */
  uint64_t path_queue_bytes;  /* device memory the context's path queues hold right now (freed by rt_resize / rt_set_partition;
                                 a launch that spills allocates what it needs, at most 1 GiB)                              */
  uint32_t dirmask_lists;     /* 0: the table holds bit sets (scenes of up to 128 spheres); otherwise it holds candidate lists
                                 (larger scenes) and this is the largest reach of an entry, in cells                    */
  uint32_t spheres_scanned;   /* sphere records the kernels hold: sphere_count minus the records that repeat an earlier one bit for
                                 bit (the zero-filled tail of the reference's 80-slot buffer: 15 -> 1), which can never win
                                 against their first occurrence (ray.hlsl:98-106: strict '<', ascending order)              */
  uint32_t last_launches_in_flight; /* launches that may overlap on the device the way the last one was issued (RT_OPT_INFLIGHT for
                                 rt_render, RT_OPT_BATCH_INFLIGHT or the library's rule for rt_render_frames; 1: one at a time)      */
  uint32_t reserved0;
} rt_stats;

/* Device-side view of the context's buffers, for zero-copy consumers (RCCL gather,
 * interop).  All pointers are device addresses on the context's GPU and stay valid
 * until the next rt_resize / rt_set_partition / rt_set_outputs / rt_destroy. */
typedef struct rt_device_buffers {
  void *rgba8;          /* local_rows x width x 4 B, sRGB-encoded, alpha 255 (with a communicator: the target the
                           last launch wrote — the two targets alternate, so ask again after every launch) */
  void *rgbf32;         /* local_rows x width x 3 floats, or NULL            */
  void *segments;       /* local_rows x width x u32, or NULL                 */
  void *hits;           /* local_rows x width x reflection_limit x i32, or NULL */
  void *stream;         /* hipStream_t all of the context's work is ordered on */
  uint32_t width, local_rows;
} rt_device_buffers;

/* `Initial::gpu_setup` (src/app.rs:64-104): bind one GPU.  One context per process
 * and GPU; multi-GPU runs create one context per rank (see rt_set_partition). */
RT355_API int rt_create(int device_id, rt_context **out_ctx);
RT355_API int rt_destroy(rt_context *ctx);
RT355_API uint32_t rt_abi_version(void);
/* 1 in the A/B build (librt355_ab.so, `make -C rsraytracingv3_amd/csrc ab`), which also holds the launch shapes of finished A/Bs
 * (RT_OPT_VARIANT 2 and 4); 0 in the shipped library, where those two values select the re-packing shape (3). */
RT355_API int rt_has_ab_shapes(void);

/* `Connection::load_hdri` (src/gpu/connection.rs:176-215): RGBA32F equirectangular
 * panorama, tightly packed rows, row 0 = top.  Sampled nearest / clamp-to-edge
 * (connection.rs:164-174). */
RT355_API int rt_set_skybox(rt_context *ctx, uint32_t width, uint32_t height, const float *rgba32f);

/* `Connection::load_image` + `load_hdri` for the reference's asset format (connection.rs:154-215; `image`
 * 0.25.1 / `exr` 1.72.0 per Cargo.lock): read an OpenEXR file — scan-line or tiled (the full-resolution
 * level), single- or multi-part (the first part that is not deep and has R, G, B), compression NONE, RLE,
 * ZIPS, ZIP, PIZ or PXR24; HALF / FLOAT / UINT channels R, G, B and optionally A — into RGBA32F, alpha 1
 * where the file has none, rows top to bottom of the display window.  Host code only (usable without a
 * GPU).  *out_rgba32f is released with rt_exr_free.  Errors: RT_ERR_INVALID_ARGUMENT with the text in
 * rt_exr_last_error() (per thread), worded like the reference's ("Can't find image: ...", "Image
 * invalid format: ...").  rt_set_skybox_exr = read + rt_set_skybox. */
RT355_API int rt_exr_read_rgba32f(const char *path, uint32_t *out_width, uint32_t *out_height, float **out_rgba32f);
RT355_API void rt_exr_free(float *rgba32f);
RT355_API const char *rt_exr_last_error(void);
RT355_API int rt_set_skybox_exr(rt_context *ctx, const char *path);

/* `RandomTexture::write` (src/gpu/random_texture.rs:78-101): R32F noise texture the
 * shader's random() indexes (src/shaders/random.hlsl:4-11).  Thread-safe against
 * every other call; a frame sees either the old or the new texture, never a mix. */
RT355_API int rt_set_random(rt_context *ctx, uint32_t width, uint32_t height, const float *r32f);

/* Device-side replacement of the reference's generator thread (`RandomTexture::generate_data` +
 * `write`, random_texture.rs:67-101): fills the texture on the GPU with
 * SplitMix64(seed) -> uniform [0,1) on the 2^-24 grid, texel k = output k+1, instead of generating on
 * the host and uploading 1.44 MB every 4 ms.  Same thread-safety as rt_set_random. */
RT355_API int rt_generate_random(rt_context *ctx, uint32_t width, uint32_t height, uint64_t seed);
/* Copy the current random texture to the host (width*height floats). */
RT355_API int rt_read_random(rt_context *ctx, float *dst, size_t bytes);

/* The dirty branch of `Connection::update_buffers` (connection.rs:373-385): objects in
 * the 112-byte wire format with an EXPLICIT count (the reference derives 80 from the
 * buffer size, inputs.hlsl:77-80 — pass 80 zero-padded slots for reference-exact
 * behaviour), lights (accepted, never read — as in the shader), Config blob.
 * Config.width/height must equal the size given to rt_resize. */
RT355_API int rt_set_scene(rt_context *ctx, const void *objects, uint32_t object_count, const void *lights,
                 uint32_t light_count, const void *config112);

/* `RenderTarget::new/resize` (src/gpu/render_target.rs:39-48,72-75): (re)allocate the
 * output and BOTH history buffers (the reference forgets the history). */
RT355_API int rt_resize(rt_context *ctx, uint32_t width, uint32_t height);

/* Tile partition for multi-GPU frames (no reference counterpart): the frame is cut
 * into strips of `strip_rows` rows dealt round-robin over `world` contexts; this
 * context renders the strips with index % world == rank into compact local buffers
 * (local row order = ascending global row).  rank 0 / world 1 = the whole frame. */
RT355_API int rt_set_partition(rt_context *ctx, uint32_t rank, uint32_t world, uint32_t strip_rows);

/* The multi-GPU frame behind the C-ABI: "the frame tile-partitions across the GPUs of one node with an RCCL gather
 * over xGMI to assemble the image".  The reference has one process, one adapter and one loop (src/app.rs:64-104,
 * 208-254); here one process drives one GPU, and the N processes of a node form a communicator:
 *   rank 0:      rt_comm_get_unique_id(id)      128 opaque bytes, handed to the other ranks by the host (any channel)
 *   every rank:  rt_comm_init(ctx, id, rank, world, strip_rows)   collective; also sets the partition
 *                (rt_set_partition) — rank r renders the row strips s with s % world == r
 *   every rank:  the same rt_set_scene / rt_set_skybox / rt_set_random / rt_resize / rt_render* sequence.  After
 *                each launch the rank's strips travel to rank 0 (grouped ncclSend / ncclRecv, each peer on its own
 *                xGMI link; no ring, no all-gather) on a communication stream of the context, beside the next
 *                launch; targets alternate between two buffers so the gather never meets the next launch's writes.
 *   rank 0:      rt_read_frame_rgba8 returns the assembled height x width frame (after the last gather).
 * RCCL is bound at run time (dlopen: $RT355_RCCL_LIB, an RCCL already loaded in the process, librccl.so.1), so a
 * single-GPU host needs none; rt_comm_init returns RT_ERR_NOT_READY when no RCCL can be loaded.  Per-pixel
 * results do not depend on the partition (the RNG state is per pixel, random.hlsl:1-2,45-47): the assembled frame
 * is bit-identical to the single-GPU frame. */
#define RT355_COMM_ID_BYTES 128
/* Can an RCCL be bound in this process?  RT_OK / RT_ERR_NOT_READY; never a collective, needs no GPU and no context.  The
 * text (the library bound, or why each candidate failed) goes to why[0..why_cap) when given.  Hosts call it on every
 * rank and agree on the outcome BEFORE any rank enters rt_comm_init: a rank that cannot load RCCL returns from
 * rt_comm_init at once while the others would wait inside ncclCommInitRank for it. */
RT355_API int rt_comm_available(char *why, size_t why_cap);
RT355_API int rt_comm_get_unique_id(void *id128);
RT355_API int rt_comm_init(rt_context *ctx, const void *id128, uint32_t rank, uint32_t world, uint32_t strip_rows);
RT355_API int rt_comm_destroy(rt_context *ctx);
/* Watchdog for the exchange: rt_comm_idle sets *idle to 1 when everything queued on the context (renders, resolves, the
 * gathers on the communication stream) has completed, 0 otherwise, without waiting.  rt_comm_abort gives up on an
 * exchange that does not complete (a peer that never posted its side): ncclCommAbort, then the context is a
 * single-GPU context again.  rt_comm_destroy and rt_destroy wait for an exchange in flight for at most 20 s and then
 * abort it the same way: no call of this interface waits forever on a peer. */
RT355_API int rt_comm_idle(rt_context *ctx, int *idle);
RT355_API int rt_comm_abort(rt_context *ctx);
/* The whole frame, sRGB8 texels, height x width x 4 bytes: on one GPU the same as rt_read_rgba8; with a
 * communicator only on rank 0 (other ranks: RT_ERR_INVALID_ARGUMENT; their rt_read_rgba8 still returns their rows). */
RT355_API int rt_read_frame_rgba8(rt_context *ctx, void *dst, size_t bytes);

/* The same multi-GPU frame driven by ONE process — the reference's shape: one process, one loop (src/app.rs:64-104 binds
 * the adapter, :208-254 renders).  rt_multi_create binds n devices (`Initial::gpu_setup` once per device) and gives device i
 * the row strips s with s % n == i; every rt_multi_set_* forwards to all devices (scene, skybox and random texture are
 * replicated), every rt_multi_render* launches on all of them — the launches are asynchronous, the devices work side by
 * side — and then gathers on device_ids[0]: peer copies of the other devices' strips (hipMemcpyPeerAsync, one xGMI link per
 * peer, no ring) into the layout rank 0 of a communicator receives, then the same assembly kernel; targets alternate, so
 * the gather of launch k runs beside launch k + 1.  rt_multi_read_frame_rgba8 returns the assembled height x width frame,
 * bit-identical to the single-GPU frame.  rt_multi_set_random / rt_multi_generate_random are thread-safe against the render
 * calls, and the texture changes on all devices between two launches: a frame sees ONE texture in every strip
 * (random_texture.rs:56-61 refreshes it from a thread of its own).  device_ids may name one device several times (what the
 * tests do on one GPU).
 * rt_multi_context(m, i) is device i's context for per-device calls (rt_get_stats, rt_set_option, rt_last_error); do not
 * render, resize or destroy through it.  No RCCL involved; use rt_comm_init when the ranks are separate processes. */
typedef struct rt_multi rt_multi;
RT355_API int rt_multi_create(const int *device_ids, uint32_t n_devices, uint32_t strip_rows, rt_multi **out);
RT355_API int rt_multi_destroy(rt_multi *m);
RT355_API uint32_t rt_multi_device_count(rt_multi *m);
/* How the gather reaches the first device: members on another device with direct peer access to it (hipDeviceEnablePeerAccess: one
 * xGMI hop per strip) and members whose copies the runtime stages (no peer access: slower, still correct).  Members on the first
 * device itself count in neither.  NOTE: this environment has one GPU per box — the cross-device path has never run on two physical
 * devices (tests bind one device several times); tests/test_multi_gpu.py holds a test that runs only where two are visible. */
RT355_API int rt_multi_peer_status(rt_multi *m, uint32_t *direct, uint32_t *staged);
RT355_API rt_context *rt_multi_context(rt_multi *m, uint32_t index);
RT355_API const char *rt_multi_last_error(rt_multi *m);
RT355_API int rt_multi_set_skybox(rt_multi *m, uint32_t width, uint32_t height, const float *rgba32f);
RT355_API int rt_multi_set_skybox_exr(rt_multi *m, const char *path);
RT355_API int rt_multi_set_random(rt_multi *m, uint32_t width, uint32_t height, const float *r32f);
RT355_API int rt_multi_generate_random(rt_multi *m, uint32_t width, uint32_t height, uint64_t seed);
RT355_API int rt_multi_set_scene(rt_multi *m, const void *objects, uint32_t object_count, const void *lights, uint32_t light_count,
                                 const void *config112);
RT355_API int rt_multi_resize(rt_multi *m, uint32_t width, uint32_t height);
RT355_API int rt_multi_set_accum_mode(rt_multi *m, int mode);
RT355_API int rt_multi_set_option(rt_multi *m, int option, int value);
RT355_API int rt_multi_render(rt_multi *m, float jitter_x, float jitter_y, uint32_t progressive_count);
RT355_API int rt_multi_render_frame(rt_multi *m, const void *framedata16);
RT355_API int rt_multi_render_frames(rt_multi *m, uint32_t n_frames, const void *framedata16);
RT355_API int rt_multi_sync(rt_multi *m);
RT355_API int rt_multi_read_frame_rgba8(rt_multi *m, void *dst, size_t bytes);
RT355_API int rt_multi_present_async(rt_multi *m, uint32_t slot); /* = rt_present_async, of the assembled frame (device 0) */
RT355_API int rt_multi_present_wait(rt_multi *m, uint32_t slot, const void **texels, size_t *bytes, uint32_t *rows, uint32_t *width);
/* Sums over the devices (segments, rough hits, pixels, local rows, wave iterations); last_kernel_ms = the slowest device's. */
RT355_API int rt_multi_get_stats(rt_multi *m, rt_stats *out);

/* Rank 0's assembly step on its own, for hosts that move the strips by another transport (and for testing the
 * layout without a second GPU): rank0_rows = rank 0's local rows, peer_rows = the local rows of ranks 1 .. world-1
 * concatenated in rank order (exactly what rank 0 receives), all host pointers to sRGB8 texels; frame gets the
 * height x width image.  Same layout function and kernel as the gather of rt_comm_init. */
RT355_API int rt_assemble_strips(rt_context *ctx, uint32_t width, uint32_t height, uint32_t world, uint32_t strip_rows,
                                 const void *rank0_rows, const void *peer_rows, void *frame);

RT355_API int rt_set_accum_mode(rt_context *ctx, int mode);
RT355_API int rt_set_outputs(rt_context *ctx, uint32_t out_flags);

/* The render pass + history copy + submit of `App::render_scene` (app.rs:219-250) and
 * the FrameData upload of update_buffers (connection.rs:390-409).  Jitter and
 * progressive_count are supplied by the caller (the reference draws them itself,
 * which makes its output irreproducible).  Asynchronous: returns after the launch. */
RT355_API int rt_render(rt_context *ctx, float jitter_x, float jitter_y, uint32_t progressive_count);
RT355_API int rt_render_frame(rt_context *ctx, const void *framedata16); /* same, FrameData blob */
/* n_frames (1..64) consecutive frames in ONE kernel launch: framedata16 = n_frames FrameData blobs
 * (connection.rs:13-33), i.e. a jitter and a progressive_count per frame.  The target, its history and
 * rt_read_* afterwards are bit-identical to n_frames rt_render_frame calls in that order with the
 * random texture unchanged in between (the launch computes every frame's float sample, a second
 * small kernel replays frag.hlsl:60-65 and the sRGB8 store frame by frame).  For progressive
 * accumulation (app.rs:208-254 calls one frame per loop turn): a launch of several frames fills the
 * GPU where one frame, or one rank's strips of it, cannot.  RT_OUT_SEGMENTS / RT_OUT_HITS describe a
 * single frame and must be off when n_frames > 1. */
RT355_API int rt_render_frames(rt_context *ctx, uint32_t n_frames, const void *framedata16);
/* Intermediate frames of a multi-frame launch.  The reference presents every frame (app.rs:240-253: copy to the
 * history, refresh the egui texture); rt_render_frames leaves only the state after its last frame in the target.
 * With rt_keep_batch_frames(ctx, every) the stored texels after frames every-1, 2*every-1, ... of each later
 * launch are kept as well (exactly what n single rt_render_frame calls would have shown at those points) and
 * rt_read_batch_frame(ctx, i, ...) reads frame i of the LAST launch (i + 1 a multiple of `every`; local rows,
 * 4 bytes per texel).  every = 0 (default) keeps nothing; every = 1 keeps all n frames (n x the target's size). */
RT355_API int rt_keep_batch_frames(rt_context *ctx, uint32_t every);
RT355_API int rt_read_batch_frame(rt_context *ctx, uint32_t frame_in_launch, void *dst, size_t bytes);
RT355_API int rt_sync(rt_context *ctx);

/* `RenderTarget::update` hands the wgpu view to egui (render_target.rs:50-70); here the
 * caller reads the pixels back (or uses rt_get_device_buffers).  Local rows only. */
RT355_API int rt_read_rgba8(rt_context *ctx, void *dst, size_t bytes);
RT355_API int rt_read_rgbf32(rt_context *ctx, float *dst, size_t bytes);
/* The same for local rows first_row .. first_row + n_rows - 1 only (a viewer that presents a region; tests
 * that follow a band of a 4K frame through an accumulation without moving 130 MB per frame). */
RT355_API int rt_read_rgba8_rows(rt_context *ctx, uint32_t first_row, uint32_t n_rows, void *dst, size_t bytes);
RT355_API int rt_read_rgbf32_rows(rt_context *ctx, uint32_t first_row, uint32_t n_rows, float *dst, size_t bytes);
RT355_API int rt_read_segments(rt_context *ctx, uint32_t *dst, size_t bytes);
RT355_API int rt_read_hits(rt_context *ctx, int32_t *dst, size_t bytes);
/* Overwrite the sRGB8 history (e.g. to resume an accumulation); local rows.  With a communicator the copy waits for the
 * gather that still sends from that buffer. */
RT355_API int rt_write_history_rgba8(rt_context *ctx, const void *src, size_t bytes);

RT355_API int rt_get_device_buffers(rt_context *ctx, rt_device_buffers *out);
RT355_API int rt_get_stats(rt_context *ctx, rt_stats *out);
/* Diagnostic: the 16 per-launch counter words of the last launch, summed over their shards — {segments, rough
 * hits, wave iterations, cull mismatches, then 8 per-phase cycle sums that only the profiling build
 * (make -C csrc prof) fills, 4 unused}. */
RT355_API int rt_get_counter_words(rt_context *ctx, uint64_t *out16);
/* Per-frame totals and HIP-event kernel times of the most recent n frames (oldest first;
 * n <= 512).  Any output pointer may be NULL.  Waits for the stream. */
RT355_API int rt_get_frame_history(rt_context *ctx, uint32_t n, uint64_t *segments, uint64_t *rough_hits,
                                   float *kernel_ms);
/* HIP-event times of the history-resolve pass that follows the render kernel of a multi-frame launch
 * (rt_render_frames), most recent n launches, oldest first; ~0 for single-frame launches. */
RT355_API int rt_get_resolve_history(rt_context *ctx, uint32_t n, float *resolve_ms);
/* HIP-event times of tail_stream, the second render kernel of a launch with a tail spill (RT_OPT_TAIL_SPILL), most recent n
 * launches, oldest first; 0 for launches without one.  It is part of the kernel time rt_get_frame_history reports. */
RT355_API int rt_get_tail_history(rt_context *ctx, uint32_t n, float *tail_ms);
/* HIP-event time from the start of the oldest to the end of the newest render kernel of the most recent n launches:
 * with several launches in flight (RT_OPT_INFLIGHT, RT_OPT_BATCH_INFLIGHT) the kernel times of rt_get_frame_history
 * overlap and their sum counts the device more than once; this span does not. */
RT355_API int rt_get_launch_span(rt_context *ctx, uint32_t n, float *span_ms);
/* Order all of the context's work on a caller-owned hipStream_t (e.g. the stream an RCCL
 * gather is issued on).  The default is a private non-blocking stream. */
RT355_API int rt_set_stream(rt_context *ctx, void *hip_stream);
/* Render into caller-owned device buffers of local_rows x width texels (rgba8: 4 B each,
 * rgbf32: 12 B each or NULL).  d_history_rgba8 is the previous frame's texels (NULL = the
 * same buffer, in place).  Pass d_rgba8 = NULL to return to the context's own buffers.
 * Takes effect at the next rt_render; does not synchronise. */
RT355_API int rt_bind_outputs(rt_context *ctx, void *d_rgba8, void *d_history_rgba8, void *d_rgbf32);
/* Text of the most recent error on this context (valid until the next call on it). */
RT355_API const char *rt_last_error(rt_context *ctx);

/* Asynchronous presentation — `RenderTarget::update` + the view egui draws every frame (src/gpu/render_target.rs:50-70,
 * src/app.rs:240-253).  The reference hands over a zero-copy view; a host that shows the frame needs the texels in host memory, and a
 * blocking rt_read_rgba8 after every frame would serialise the copy (8.3 MB at 1080p) with the rendering.  rt_present_async queues
 * a copy of the texels of the frames launched so far — local_rows x width x 4 bytes; on rank 0 of a communicator (and for an rt_multi)
 * the assembled height x width frame — into pinned staging buffer `slot` (0 .. RT355_PRESENT_SLOTS - 1, owned by the context) on a
 * copy stream of the context, and returns at once: the frame is snapshotted on the device (the target is also the history, which the
 * next launch overwrites in place) and travels to the host beside the next launches' rendering, consecutive frames' copies back to
 * back at the link's rate (53 GB/s measured: 0.16 ms per 1080p frame).  rt_present_wait blocks until that copy has arrived and returns the buffer,
 * valid until the slot's next rt_present_async or the next rt_resize / rt_set_partition / rt_destroy.  Alternate two slots to
 * show frame k while frame k + 1 is copied.  The history is untouched; frames equal the synchronous read-backs texel for texel. */
#define RT355_PRESENT_SLOTS 4
RT355_API int rt_present_async(rt_context *ctx, uint32_t slot);
RT355_API int rt_present_wait(rt_context *ctx, uint32_t slot, const void **texels, size_t *bytes, uint32_t *rows, uint32_t *width);

/* Tuning knobs; results are bit-identical for every setting, only speed changes. */
enum {
  RT_OPT_VARIANT = 0,      /* -1 (default) choose per launch; 0 one 8x8 tile per wave, sphere records in
                              LDS; 1 same, records from global memory; 3 tiles whose workgroup re-packs surviving paths.
                              Only in the A/B build (rt_has_ab_shapes; the shipped library runs 3 for both): 2 resident waves that
                              refill finished lanes; 4 streamed (sample-only launches: primary pass -> path queue in global
                              memory -> resident waves refilled from the queue after every bounce; else falls back to 3)  */
  RT_OPT_REFILL_MIN = 1,   /* variants 2 and 4: idle lanes a wave tolerates before pulling work (1..64) */
  RT_OPT_BLOCKS_PER_CU = 2,/* variants 2 and 4: resident workgroups per CU (0 = occupancy query) */
  RT_OPT_TILE_CHUNK = 3,   /* variant 2: consecutive 8x8 tiles a wave takes per work-counter bump; variant 4: 64-path
                              units a wave takes off the path queue at a time */
  RT_OPT_FLAGS = 4,        /* experiment switches: bit 1 (2) ignore the sphere grid; bit 2 (4) launches of every size spill their
                              tail (RT_OPT_TAIL_SPILL) — speed only, frames stay bit-identical; bit 3 (8) rt_set_scene keeps sphere records that repeat an
                              earlier one (rt_stats.spheres_scanned) — speed only; bit 5 (32) spheres far from all the others size the candidate
                              tables like any other instead of being entered into every set (rt_dirmask_info.n_outliers) — speed only; bit 4 (16) DRY launches: every
                              kernel of a launch (render, tail, resolve, assembly) runs ONE workgroup and peer copies move one row —
                              all of a frame's host calls, events and stream waits with next to none of its device work, to tell
                              host time from device time (tools/multi_times.py); the frames are garbage                         */
  RT_OPT_CULL = 5,         /* primary-ray tile culling: 0 off, 1 on (default), 2 on + every culled
                              scan is re-done in full and disagreements are counted in rt_stats     */
  RT_OPT_INFLIGHT = 6,     /* single-frame launches (rt_render / rt_render_frame) in flight at once, 1..8, or 0 (default) = the library's rule: four,
                              five for short launches (reflection_limit <= 2: +9 % / +2 %; a rank's share of a tile-split frame,
                              i.e. fewer 8x8 tiles than a 1080p frame has: +5 ... +19 %):
                              frame k + 1 starts rendering on another internal stream while the last
                              workgroups of frame k drain (3 and 6 are slower than 2 and 5: the runtime deals streams to four
                              hardware queues in turn; the environment variable RT355_STREAM_PAD=k, a diagnostic, makes rt_create
                              create k idle streams first and so rotates that deal) — the reference's loop issues one frame per turn
                              (app.rs:208-254) and one 1080p frame leaves a fifth of the GPU idle during its
                              ramp-up and tail; the history is still updated strictly in frame order on the
                              context's stream.  1 = one launch at a time (kernel times do not overlap).     */
  RT_OPT_GRID_MIN = 7,     /* scenes with at least this many spheres (default 129) get a uniform grid at the next
                              rt_set_scene: secondary rays then test the spheres of the cells they cross instead of
                              every sphere (ray.hlsl:97-107 scans linearly; the reference caps scenes at 80)       */
  RT_OPT_DIRMASK = 9,      /* 1 (default): scenes of RT_OPT_DIRMASK_MIN .. 128 spheres that get no grid receive a table
                              of candidate-sphere sets at the next rt_set_scene — [cell of the ray's origin][cube-map bin of
                              its direction] -> set of sphere slots — and secondary rays test only their set instead of every
                              sphere (ray.hlsl:97-107 scans linearly); rebuilt only when the sphere records change; 0: off */
  RT_OPT_DIRMASK_MIN = 10,  /* fewest spheres that get the table (default 26)                                            */
  RT_OPT_DIRMASK_CELLS = 11,/* cells the table's grid aims at (default 4096; cubic cells, at most 64 per axis)           */
  RT_OPT_DIRMASK_K = 12,    /* direction bins per cube-map face edge, 1..16 (default 12: 864 bins)                       */
  RT_OPT_TAIL_SPILL_SINGLE = 14, /* the same threshold for single-frame launches in flight (rt_render with RT_OPT_INFLIGHT >= 2 and no
                             * diagnostic outputs) of 24 000 tiles or more (a whole 1080p frame); default 32 (0 = off): tail_stream costs a
                             * single frame ~35 us whatever it holds, which pays in an uninterrupted loop of whole frames (+3.6 % at
                             * 1080p and 8 bounces) and not for a rank's share of a frame or between synchronisations            */
  RT_OPT_TAIL_SPILL = 13,   /* multi-frame launches (rt_render_frames) of the re-packing shape, of 100 000 tiles or more (three
                             * 1080p frames) and two bounces or more, on a scene with a direction-mask table: a workgroup whose paths that hit number at most
                             * this many (0..128; -1 = the default: 32, 24 on a scene with candidate lists; 0 = off) hands them — path state + pending hit, 80 B — to a path
                             * queue in device memory and ends; a second kernel, tail_stream, takes them up into full waves for
                             * their remaining bounces instead of letting each workgroup bounce its last few paths on in a
                             * mostly empty wave (bit-identical; +5 % at 1080p, +9 % at 3840x2160).  The queue holds at most
                             * workgroups x threshold records (0.33 GB for 8 frames of 1080p); above 1 GiB the threshold shrinks; rt_stats reports the
                             * queue's size and a launch that had to go without it */
  RT_OPT_DIRLIST = 15,      /* 1 (default): scenes of RT_OPT_DIRLIST_MIN spheres or more (up to 65 535) receive a table of candidate
                             * LISTS at the next rt_set_scene instead of the sphere grid: [cell of the point a ray has reached][bin of
                             * its direction] -> up to 15 sphere slots + the distance up to which the list is complete (its reach);
                             * a ray tests the list and looks the next one up further along when its best hit lies beyond the
                             * reach.  Such scenes then render in the re-packing shape (3) like the small ones, nothing staged into
                             * LDS.  0: off (the grid of RT_OPT_GRID_MIN serves).  RT_OPT_DIRMASK = 0 switches both tables off.      */
  RT_OPT_DIRLIST_MIN = 16,  /* fewest spheres that get candidate lists (default 129: the bit-set table serves up to 128)          */
  RT_OPT_DIRLIST_CELLS = 17,/* cells the lists' grid aims at (default 16384; cubic cells, at most 128 per axis)                  */
  RT_OPT_DIRLIST_K = 18,    /* their direction bins per cube-map face edge, 1..16 (default 10: 600 bins)                           */
  RT_OPT_DIRLIST_REACH = 19,/* the largest reach of an entry, in cells (default 32)                                             */
  RT_OPT_TIMING = 20,       /* 1 (default): every launch records the HIP events rt_get_frame_history / rt_get_stats report kernel times from;
                             * 0: single-frame launches in flight record none that only serve those queries (their kernel times read 0):
                             * a frame then costs the host six HIP calls, 16 us instead of 27 (tools/microbench/host_calls.hip) — what a
                             * render loop that never asks for kernel times wants when a rank's share of a frame is a few microseconds */
  RT_OPT_BATCH_INFLIGHT = 8 /* multi-frame launches (rt_render_frames) in flight at once, 1..8, or 0 (default) = the library's
                              rule: four while the launches' sample buffers fit 2 GiB together, else down to two
                              (rt_stats.last_launches_in_flight says what a launch got).  As RT_OPT_INFLIGHT, each launch
                              stages its samples in a buffer of its own; the history is updated in frame order on the
                              context's stream.  Launches in flight hide the resolve pass, the tail of a launch and the gap to
                              the next one behind the next launches' rendering (1 -> 2: +4 ... +8 % throughput at 1080p, 2 -> 4:
                              +1 ... +5 %) at the price of the staging buffers and of kernel times that overlap
                              (rt_get_launch_span gives the device time); 1 = one launch at a time.                          */
};
RT355_API int rt_set_option(rt_context *ctx, int option, int value);
RT355_API int rt_set_variant(rt_context *ctx, int variant); /* = rt_set_option(RT_OPT_VARIANT) */

/* The sphere grid rt_set_scene builds for large scenes (RT_OPT_GRID_MIN), exposed for inspection and for testing its
 * registration rule without a GPU: host code only.  spheres = n x {cx, cy, cz, r*r}; cell_start gets n_cells + 1
 * entries, items n_items sphere indices (when the capacities given suffice; call once with null arrays for the
 * sizes).  info->enabled = 0 when no grid is built (too few spheres, non-finite data, or too many entries). */
typedef struct rt_grid_info {
  uint32_t enabled, nx, ny, nz, n_cells, n_items;
  float gmin[3], gmax[3], cell[3], inv_cell[3], near_min[3], near_max[3], t_margin;
} rt_grid_info;
RT355_API int rt_grid_build_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t min_spheres, rt_grid_info *info,
                                 uint32_t *cell_start, uint32_t cell_cap, uint16_t *items, uint32_t item_cap);

/* The direction-mask table rt_set_scene builds for scenes of up to 128 spheres (RT_OPT_DIRMASK), exposed for inspection
 * and for testing its sets without a GPU: host code only, the same entry function the device kernel runs.  spheres =
 * n x {cx, cy, cz, r*r}; table gets nx*ny*nz*bins sets of `words` 64-bit words (1 for up to 64 spheres, else 2),
 * [cell][bin][word] (when table_cap, in words, suffices; call with a null table for the sizes).  info->enabled = 0 when the
 * scene is not served.  rt_dirmask_lookup_host repeats a lane's lookup: *served = 0 when the ray is not served by the table
 * (origin outside the near box, direction not a unit vector — the kernels then scan linearly), else mask2[0..1] = the ray's
 * candidate set (all zero for a ray that misses the spheres' box). */
typedef struct rt_dirmask_info {
  uint32_t enabled, nx, ny, nz, k, bins, words;
  float bmin[3], bmax[3], cell[3], inv_cell[3], near_min[3], near_max[3];
  uint32_t n_outliers, outlier_slot[4]; /* spheres far from all the others: they do not size the table (boxes, cells, rounding term)
                                           and are members of every set instead                                              */
} rt_dirmask_info;
RT355_API int rt_dirmask_build_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t target_cells, uint32_t k,
                                    rt_dirmask_info *info, uint64_t *table, size_t table_cap);
RT355_API int rt_dirmask_lookup_host(const rt_dirmask_info *info, const uint64_t *table, const float *origin3,
                                     const float *direction3, uint64_t *mask2, int *served);

/* The candidate lists rt_set_scene builds for scenes of more than 128 spheres (RT_OPT_DIRLIST), exposed for testing them without a
 * GPU: host code only, the same plan and entry functions the device builder runs.  rt_dirlist_plan_host: the table's geometry and,
 * in registered_xyzr (n x 4 floats, may be null), the registered balls {c, r_reg}.  rt_dirlist_entry_host: one entry, 8 words =
 * 16 half-words: h[0] = n << 12 | reach quanta, h[1..n] = sphere slots (a whole table is hundreds of megabytes; the device builds
 * it).  rt_dirlist_walk_host: one ray through a lane's rounds, entries computed on demand, next to the linear scan of every sphere
 * in the same arithmetic — the two (key, slot) results must be equal for every ray; *far = 1 when the ray is not served by the table
 * (the kernels then scan linearly), t_plane = the distance of the ray's plane hit (it ends the rounds like a sphere hit), rounds /
 * tests = lookups made / spheres tested. */
typedef struct rt_dirlist_info {
  uint32_t enabled, nx, ny, nz, k, bins;
  float bmin[3], bmax[3], cell[3], inv_cell[3], reach_unit, s_cap;
  uint32_t n_outliers, outlier_slot[4]; /* spheres far from all the others: registered with radius -1, they do not size the table
                                           and are listed by every entry instead                                                */
} rt_dirlist_info;
RT355_API int rt_dirlist_plan_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t target_cells, uint32_t k,
                                   uint32_t reach_cells, rt_dirlist_info *info, float *registered_xyzr);
RT355_API int rt_dirlist_entry_host(const rt_dirlist_info *info, const float *registered_xyzr, uint32_t n_spheres, uint32_t cell,
                                    uint32_t bin, uint32_t *entry8);
RT355_API int rt_dirlist_walk_host(const rt_dirlist_info *info, const float *spheres_xyzr2, const float *registered_xyzr,
                                   uint32_t n_spheres, const float *origin3, const float *direction3, float t_plane,
                                   uint32_t *listed_key, int32_t *listed_slot, uint32_t *linear_key, int32_t *linear_slot,
                                   uint32_t *rounds, uint32_t *tests, int *far);

/* Device self-test hooks used by the parity tests: evaluate the device build of the
 * specified math function over n host values (op: 0 sin 1 cos 2 tan 3 asin 4 log2
 * 5 exp2 6 sqrt 7 the kernel's own sqrt; binary op: 0 atan2(a,b) 1 pow(a,b) 2 a/b). */
RT355_API int rt_selftest_sqrt(rt_context *ctx, uint64_t *mismatches); /* kernel sqrt vs IEEE, every float */
RT355_API int rt_selftest_math_unary(rt_context *ctx, int op, const float *x, float *y, size_t n);
RT355_API int rt_selftest_math_binary(rt_context *ctx, int op, const float *a, const float *b, float *y, size_t n);

#ifdef __cplusplus
}
#endif

#endif /* RT355_H */

/* rt_oracle.c — CPU restatement of the reference's per-pixel ray tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rsraytracingv3_amd/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / reported baseline.
 *
 * PARITY STATUS: pinned on geometry and emission by the one output the reference
 * holds, texel values otherwise "parity unpinned".  The reference
 * (arlohb/RSRayTracingV3 @ v2) has no tests, golden images or known-answer
 * vectors, and it can be neither compiled nor run in this environment (no Rust
 * toolchain, no shader compiler, no Vulkan).  Its screenshot,
 * however, contains the unscaled 1000x900 texels its shader stored
 * (tests/golden/make_screenshot_fixture.py): tests/test_screenshot_pin.py checks
 * this restatement against them — horizon row, registration of every sphere
 * silhouette to +-1..2 px, saturation of every emissive sphere in exactly its
 * emission channels, and the colour of the light on the ground and of every
 * sphere against this restatement's own accumulation (chroma correlation 0.89 /
 * 0.92).  The HDRI, jitter and noise texture of that run are unknown,
 * so colours cannot be compared texel by texel.  Beyond that the restatement
 * follows the HLSL text function by function (citations below are relative to
 * /root/reference/src/) and is pinned by hand-derivable known-answer tests
 * (tests/test_oracle_kat.py) and byte-layout KATs derived from the Rust writers
 * (tests/test_layout.py).
 *
 * Arithmetic: IEEE-754 binary32 throughout, built with -ffp-contract=off; every
 * fused multiply-add is an explicit fmaf().  Where the HLSL leaves a choice
 * (contraction, operand order, intrinsic precision) the choice is written once
 * in DESIGN.md ("Arithmetic specification", decisions D1-D18) and repeated in
 * the comment of the function that makes it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "rt_oracle_math.h"

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* a*s + b, one rounding per component (D15: products feeding a sum are fused) */
static inline v3 v_madd(v3 a, float s, v3 b) {
  return V(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z));
}
/* D12/D15: dot accumulates x, y, z in that order, the y and z steps fused */
static inline float v_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* D12/D15: cross(a,b).x = a.y*b.z - a.z*b.y with the first product fused */
static inline v3 v_cross(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float v_length(v3 a) { return sqrtf(v_dot(a, a)); }
/* D12: normalize(v) = v * (1 / sqrt(dot(v,v))) — one IEEE sqrt, one IEEE divide */
static inline v3 v_normalize(v3 a) { return v_scale(a, 1.0f / sqrtf(v_dot(a, a))); }
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* ------------------------------------------------- wire layouts (App. A) */
/* Offsets follow the Rust writers: ray_tracer/objects.rs:41-57 (Material 48 B),
 * :125-157 (Geometry 64 B), :202-214 (Object 112 B); ray_tracer/scene.rs:252-271
 * (Config 112 B); gpu/connection.rs:30-40 (FrameData 16 B); and the HLSL
 * declarations shaders/inputs.hlsl:1-55. */
enum { OBJECT_STRIDE = 112, CONFIG_SIZE = 112, FRAMEDATA_SIZE = 16 };

typedef struct {
  v3 colour;               /* +0  */
  v3 emission;             /* +16 */
  float emission_strength; /* +28 */
  float metallic;          /* +32 (never read by the shader) */
  float roughness;         /* +36 */
} Material;

typedef struct {
  uint32_t option; /* +48: 0 sphere, 1 plane */
  v3 center;       /* +64 */
  v3 vec_data;     /* +80: plane normal */
  float f32_data;  /* +92: sphere radius / plane half extent */
} Geometry;

typedef struct { Material material; Geometry geometry; } Object;

typedef struct {
  v3 position, forward, right, up; /* +0, +16, +32, +48 */
  float fov;                       /* +92, degrees */
  uint32_t reflection_limit;       /* +96 */
  uint32_t width, height;          /* +100, +104 */
} Config;

typedef struct { float jitter_x, jitter_y; uint32_t progressive_count; } FrameData;

static float rd_f32(const uint8_t *p) { float f; memcpy(&f, p, 4); return f; }
static uint32_t rd_u32(const uint8_t *p) { uint32_t u; memcpy(&u, p, 4); return u; }
static v3 rd_v3(const uint8_t *p) { return V(rd_f32(p), rd_f32(p + 4), rd_f32(p + 8)); }

static Object parse_object(const uint8_t *p) {
  Object o;
  o.material.colour = rd_v3(p + 0);
  o.material.emission = rd_v3(p + 16);
  o.material.emission_strength = rd_f32(p + 28);
  o.material.metallic = rd_f32(p + 32);
  o.material.roughness = rd_f32(p + 36);
  o.geometry.option = rd_u32(p + 48);
  o.geometry.center = rd_v3(p + 64);
  o.geometry.vec_data = rd_v3(p + 80);
  o.geometry.f32_data = rd_f32(p + 92);
  return o;
}

static Config parse_config(const uint8_t *p) {
  Config c;
  c.position = rd_v3(p + 0);
  c.forward = rd_v3(p + 16);
  c.right = rd_v3(p + 32);
  c.up = rd_v3(p + 48);
  c.fov = rd_f32(p + 92);
  c.reflection_limit = rd_u32(p + 96);
  c.width = rd_u32(p + 100);
  c.height = rd_u32(p + 104);
  return c;
}

static FrameData parse_framedata(const uint8_t *p) {
  FrameData f;
  f.jitter_x = rd_f32(p + 0);
  f.jitter_y = rd_f32(p + 4);
  f.progressive_count = rd_u32(p + 8);
  return f;
}

/* ------------------------------------------------------------ bound inputs */
/* shaders/inputs.hlsl:66-73: objects (b0), config (b2), sampler (b3: nearest,
 * clamp-to-edge — gpu/connection.rs:164-174), t_hdri (b4, Rgba32Float),
 * t_render (b5, Rgba8UnormSrgb history), t_random (b6, R32Float),
 * frame_data (b7).  lights (b1) are never read. */
typedef struct {
  const Object *objects;
  uint32_t object_count; /* inputs.hlsl:75-80: buffer size / 112 (80 in the reference) */
  Config config;
  FrameData frame_data;
  const float *hdri; uint32_t hdri_w, hdri_h;     /* RGBA32F rows, row 0 = top */
  const float *random; uint32_t random_w, random_h; /* R32F */
  /* per-frame camera values of create_ray that do not depend on the pixel */
  v3 top_left; float vp_width, vp_height;
} Scene;

/* Nearest + clamp-to-edge addressing: texel = clamp(floor(u*size), 0, size-1).
 * fmaxf/fminf drop a NaN operand, so a NaN coordinate addresses texel 0. */
static inline uint32_t texel_index(float u, uint32_t size) {
  float t = floorf(u * (float)size);
  t = fminf(fmaxf(t, 0.0f), (float)(size - 1));
  return (uint32_t)t;
}

/* ------------------------------------------------------------- random.hlsl */
typedef struct { float seed; float px, py; } Rng; /* random.hlsl:1-2 */

/* random.hlsl:45-47 */
static inline void random_init(Rng *rng, float px, float py) { rng->seed = 42.0f; rng->px = px; rng->py = py; }

/* random.hlsl:4-11.  coord = (rng_pixel + rng_seed) % 1.0 (D4: operands are
 * non-negative, so % 1.0 is x - floor(x)); both SampleLevel calls address the
 * same texel; rng_seed += x * 532.3412 is one fused step (D15); % 600 is the
 * exact fmodf. */
static inline float rng_random(const Scene *sc, Rng *rng) {
  float sx = rng->px + rng->seed, sy = rng->py + rng->seed;
  float cx = sx - floorf(sx), cy = sy - floorf(sy);
  uint32_t i = texel_index(cx, sc->random_w), j = texel_index(cy, sc->random_h);
  float x = sc->random[(size_t)j * sc->random_w + i];
  rng->seed = fmaf(x, 532.3412f, rng->seed);
  rng->seed = fmodf(rng->seed, 600.0f);
  return x;
}

/* random.hlsl:13-23.  Returns the three basis vectors; D2: the float3x3 built
 * from (tangent, binormal, normal) multiplies as tangent*x + binormal*y + normal*z. */
static inline void get_tangent_space(v3 normal, v3 *tangent, v3 *binormal) {
  v3 helper = V(1.0f, 0.0f, 0.0f);
  if (fabsf(normal.x) > 0.99f) helper = V(0.0f, 0.0f, 1.0f);
  *tangent = v_normalize(v_cross(normal, helper));
  *binormal = v_normalize(v_cross(normal, *tangent));
}

/* random.hlsl:25-43 */
static inline v3 random_in_hemisphere(const Scene *sc, Rng *rng, v3 normal, float roughness) {
  if (roughness == 0.0f) return normal;
  float smoothness = 1.0f - roughness;
  float phong_alpha = rto_pow(1000.0f, smoothness * smoothness);
  float cos_theta = rto_pow(rng_random(sc, rng), 1.0f / (phong_alpha + 1.0f));
  float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
  float phi = RTO_TWO_PI * rng_random(sc, rng);
  float sphi, cphi;
  rto_sincos(phi, &sphi, &cphi);
  v3 t = V(cphi * sin_theta, sphi * sin_theta, cos_theta);
  v3 tangent, binormal;
  get_tangent_space(normal, &tangent, &binormal);
  /* D2 + D15: tangent*t.x, then binormal*t.y and normal*t.z fused in */
  return v_madd(normal, t.z, v_madd(binormal, t.y, v_scale(tangent, t.x)));
}

/* --------------------------------------------------------------- ray.hlsl */
typedef struct { v3 origin, direction, energy; } Ray;                          /* ray.hlsl:1-5 */
typedef struct { v3 position; float distance; v3 normal; int object_index; } Hit; /* ray.hlsl:7-12 */

/* The pixel-independent part of create_ray (ray.hlsl:16-29), evaluated once
 * per frame.  fov*PI/180 uses D16 (constant divisor -> reciprocal multiply);
 * "/ 2." is an exact halving. */
static void camera_setup(Scene *sc) {
  const Config *c = &sc->config;
  float fov_rad = (c->fov * RTO_PI) * 0.0055555557f; /* 1/180 */
  float half_width = rto_tan(fov_rad * 0.5f);
  float half_height = (half_width * (float)c->height) / (float)c->width;
  v3 center = v_sub(c->position, c->forward); /* D9: rays leave along -forward */
  v3 left = v_madd(c->right, -half_width, center);
  v3 top = v_madd(c->up, half_height, center);
  sc->top_left = v_sub(v_add(left, top), center);
  sc->vp_width = half_width * 2.0f;
  sc->vp_height = half_height * 2.0f;
}

/* ray.hlsl:31-37: pixel_world_space = top_left + right*(u*width) - up*(v*height) */
static inline Ray create_ray(const Scene *sc, float u, float v) {
  const Config *c = &sc->config;
  v3 p = v_madd(c->right, u * sc->vp_width, sc->top_left);
  p = v_madd(v_neg(c->up), v * sc->vp_height, p);
  Ray r;
  r.origin = c->position;
  r.direction = v_normalize(v_sub(p, c->position));
  r.energy = V(1.0f, 1.0f, 1.0f);
  return r;
}

/* utils.hlsl:5-16.  D3: pow(b, 2.) is b*b; D15: b*b - 4ac is one fma. */
static inline float solve_quadratic(float a, float b, float c) {
  float discriminant = fmaf(b, b, -((4.0f * a) * c));
  if (discriminant < 0.0f) return 10000000.0f;
  return (-b - sqrtf(discriminant)) / (2.0f * a);
}

/* ray.hlsl:40-48 */
static inline v3 object_normal(const Object *o, v3 position) {
  if (o->geometry.option == 0) return v_normalize(v_sub(position, o->geometry.center));
  if (o->geometry.option == 1) return o->geometry.vec_data;
  return V(0.0f, 1.0f, 0.0f);
}

/* ray.hlsl:50-95 */
static inline void object_intersect(const Scene *sc, Hit *hit, uint32_t i, const Ray *ray) {
  const Object *o = &sc->objects[i];
  if (o->geometry.option == 0) {
    v3 new_origin = v_sub(ray->origin, o->geometry.center);
    float a = 1.0f;
    float b = 2.0f * v_dot(ray->direction, new_origin);
    float radius = o->geometry.f32_data;
    float c = v_dot(new_origin, new_origin) - radius * radius; /* D3: pow(r, 2.) = r*r, rounded */
    float distance = solve_quadratic(a, b, c);
    if (distance < hit->distance) {
      if (distance < RTO_EPSILON) return;
      hit->distance = distance;
      hit->position = v_madd(ray->direction, distance, ray->origin);
      hit->object_index = (int)i;
    }
  } else if (o->geometry.option == 1) {
    float denominator = v_dot(ray->direction, o->geometry.vec_data);
    if (fabsf(denominator) < RTO_EPSILON) return;
    float numerator = v_dot(v_sub(o->geometry.center, ray->origin), o->geometry.vec_data);
    float distance = numerator / denominator;
    v3 hit_point = v_madd(ray->direction, distance, ray->origin);
    float size = o->geometry.f32_data;
    if ((fabsf(hit_point.x - o->geometry.center.x) > size) |
        (fabsf(hit_point.y - o->geometry.center.y) > size) |
        (fabsf(hit_point.z - o->geometry.center.z) > size)) {
      return;
    }
    if (distance < hit->distance) {
      if (distance < RTO_EPSILON) return;
      hit->distance = distance;
      hit->position = hit_point;
      hit->object_index = (int)i;
    }
  }
}

/* ray.hlsl:97-107.  D6: objects[-1] is never dereferenced; the normal of a
 * miss is unused by shade(). */
static inline Hit ray_intersect(const Scene *sc, const Ray *ray) {
  Hit hit;
  hit.position = V(0.0f, 0.0f, 0.0f);
  hit.distance = 1000000.0f;
  hit.normal = V(0.0f, 0.0f, 0.0f);
  hit.object_index = -1;
  for (uint32_t i = 0; i < sc->object_count; i += 1) object_intersect(sc, &hit, i, ray);
  if (hit.object_index >= 0) hit.normal = object_normal(&sc->objects[hit.object_index], hit.position);
  return hit;
}

/* -------------------------------------------------------------- frag.hlsl */
/* frag.hlsl:8-12: equirectangular lookup, nearest texel.  D16: /(2*PI) and /PI
 * are reciprocal multiplies, fused with the +0.5 (D15). */
static inline v3 sample_hdri(const Scene *sc, v3 d) {
  float u = fmaf(rto_atan2(d.x, d.z), RTO_INV_TWO_PI, 0.5f);
  float v = fmaf(rto_asin(-d.y), RTO_INV_PI, 0.5f);
  uint32_t i = texel_index(u, sc->hdri_w), j = texel_index(v, sc->hdri_h);
  const float *t = sc->hdri + 4 * ((size_t)j * sc->hdri_w + i);
  return V(t[0], t[1], t[2]);
}

/* frag.hlsl:6-30 */
static inline v3 shade(const Scene *sc, Rng *rng, Ray *ray, const Hit *hit) {
  v3 hdri = sample_hdri(sc, ray->direction);
  if (hit->object_index == -1) {
    ray->energy = V(0.0f, 0.0f, 0.0f);
    return hdri;
  }
  const Material *m = &sc->objects[hit->object_index].material;
  ray->origin = v_madd(hit->normal, 0.001f, hit->position);
  /* reflect(i, n) = i - 2*dot(n,i)*n (D12), the subtraction fused (D15) */
  float k = 2.0f * v_dot(hit->normal, ray->direction);
  v3 reflection_ray = v_madd(hit->normal, -k, ray->direction);
  ray->direction = random_in_hemisphere(sc, rng, reflection_ray, m->roughness);
  float cosine = clamp01(v_dot(hit->normal, ray->direction));
  ray->energy = v_mul(ray->energy, v_scale(v_scale(m->colour, 2.0f), cosine));
  return v_scale(m->emission, m->emission_strength);
}

/* frag.hlsl:32-46.  D1: the left operand ray.energy is read BEFORE shade()
 * updates it; result += energy*shade is fused (D15). */
static inline v3 trace_ray_with_reflections(const Scene *sc, Rng *rng, Ray ray, uint32_t *segments, int32_t *hits,
                                            float *rays) {
  v3 result = V(0.0f, 0.0f, 0.0f);
  uint32_t n = 0;
  for (uint32_t i = 0; i < sc->config.reflection_limit; i += 1) {
    if (rays) { /* diagnostic (rto_set_ray_dump): the ray this iteration intersects */
      rays[6 * i] = ray.origin.x; rays[6 * i + 1] = ray.origin.y; rays[6 * i + 2] = ray.origin.z;
      rays[6 * i + 3] = ray.direction.x; rays[6 * i + 4] = ray.direction.y; rays[6 * i + 5] = ray.direction.z;
    }
    Hit hit = ray_intersect(sc, &ray);
    if (hits) hits[i] = hit.object_index;
    v3 energy_before = ray.energy;
    v3 s = shade(sc, rng, &ray, &hit);
    result = V(fmaf(energy_before.x, s.x, result.x), fmaf(energy_before.y, s.y, result.y),
               fmaf(energy_before.z, s.z, result.z));
    n += 1;
    if (v_length(ray.energy) < RTO_EPSILON) break;
  }
  if (segments) *segments = n;
  return result;
}

/* sRGB transfer functions of the Rgba8UnormSrgb target / history texture
 * (gpu/render_target.rs:23, app.rs:170; App. B D13). */
static inline uint8_t srgb8_encode(float c) {
  c = clamp01(c); /* NaN -> 0 */
  float s = (c <= 0.0031308f) ? 12.92f * c : fmaf(1.055f, rto_pow(c, 0.41666666f), -0.055f);
  return (uint8_t)floorf(fmaf(255.0f, s, 0.5f));
}
static inline float srgb8_decode(uint32_t b) {
  float s = (float)b / 255.0f;
  return (s <= 0.04045f) ? s / 12.92f : rto_pow((s + 0.055f) / 1.055f, 2.4f);
}

/* frag.hlsl:48-66 for one pixel.  D14: SV_POSITION.xy = (x + 0.5, y + 0.5).
 * prev = linear value of the history texel of this same pixel (D13). */
static inline v3 fs_main(const Scene *sc, uint32_t x, uint32_t y, v3 previous, uint32_t *segments, int32_t *hits,
                         float *rays) {
  float pos_x = (float)x + 0.5f, pos_y = (float)y + 0.5f;
  float pjx = pos_x + sc->frame_data.jitter_x, pjy = pos_y + sc->frame_data.jitter_y;
  float cjx = pjx / (float)sc->config.width, cjy = pjy / (float)sc->config.height;
  Rng rng;
  random_init(&rng, cjx, cjy);
  Ray ray = create_ray(sc, cjx, cjy);
  v3 colour = trace_ray_with_reflections(sc, &rng, ray, segments, hits, rays);
  float opacity = 1.0f / (float)(sc->frame_data.progressive_count + 1u);
  float keep = 1.0f - opacity;
  return V(fmaf(colour.x, opacity, previous.x * keep), fmaf(colour.y, opacity, previous.y * keep),
           fmaf(colour.z, opacity, previous.z * keep));
}

/* ------------------------------------------------------------ entry points */
#define RTO_EXPORT __attribute__((visibility("default")))

/* Diagnostic: while set, rto_render also stores the ray of every executed iteration — W x H x reflection_limit x
 * {origin.xyz, direction.xyz}, untouched where an iteration did not run — for tests of the product's culling
 * structures (every sphere a segment hits must be in the candidate set its ray is given).  NULL switches it off. */
static float *g_ray_dump = NULL;
RTO_EXPORT void rto_set_ray_dump(float *rays) { g_ray_dump = rays; }

/* Render rows [y0, y1) of one frame.
 *   objects        object_count x 112 B wire records   config112 / framedata16  wire blobs
 *   hdri           hdri_w x hdri_h RGBA32F            random     random_w x random_h R32F
 *   history_rgba8  W x H x 4 sRGB8 of the previous frame, or NULL (treated as 0)
 *   history_f32    W x H x 3 linear float history for accum_mode 1, or NULL
 *   accum_mode     0 = reference feedback through sRGB8 (D7), 1 = float32 history
 * Outputs (any may be NULL), all W x H frame-sized, rows outside [y0,y1) untouched:
 *   out_rgbf32     W x H x 3 linear "mixed" value before the sRGB store
 *   out_rgba8      W x H x 4 the stored texel
 *   out_segments   W x H     executed iterations of frag.hlsl:36-43
 *   out_hits       W x H x reflection_limit   object_index per iteration, -2 = not executed
 * Returns 0, or a negative code for invalid arguments. */
RTO_EXPORT int rto_render(const uint8_t *objects, uint32_t object_count, const uint8_t *config112,
                          const uint8_t *framedata16, const float *hdri, uint32_t hdri_w, uint32_t hdri_h,
                          const float *random, uint32_t random_w, uint32_t random_h,
                          const uint8_t *history_rgba8, const float *history_f32, int accum_mode,
                          float *out_rgbf32, uint8_t *out_rgba8, uint32_t *out_segments, int32_t *out_hits,
                          uint32_t y0, uint32_t y1, int threads) {
  if (!config112 || !framedata16 || !hdri || !random || (!objects && object_count)) return -1;
  if (!hdri_w || !hdri_h || !random_w || !random_h) return -2;
  Scene sc;
  memset(&sc, 0, sizeof sc);
  sc.config = parse_config(config112);
  sc.frame_data = parse_framedata(framedata16);
  if (!sc.config.width || !sc.config.height) return -3;
  if (y1 > sc.config.height) y1 = sc.config.height;
  Object *objs = (Object *)malloc(sizeof(Object) * (object_count ? object_count : 1));
  if (!objs) return -4;
  for (uint32_t i = 0; i < object_count; ++i) objs[i] = parse_object(objects + (size_t)i * OBJECT_STRIDE);
  sc.objects = objs;
  sc.object_count = object_count;
  sc.hdri = hdri; sc.hdri_w = hdri_w; sc.hdri_h = hdri_h;
  sc.random = random; sc.random_w = random_w; sc.random_h = random_h;
  camera_setup(&sc);
  float lut[256];
  for (uint32_t b = 0; b < 256; ++b) lut[b] = srgb8_decode(b);
  const uint32_t W = sc.config.width, R = sc.config.reflection_limit;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 2) num_threads(threads)
#endif
  for (int64_t yy = (int64_t)y0; yy < (int64_t)y1; ++yy) {
    uint32_t y = (uint32_t)yy;
    for (uint32_t x = 0; x < W; ++x) {
      size_t p = (size_t)y * W + x;
      v3 previous = V(0.0f, 0.0f, 0.0f);
      if (accum_mode == 1) {
        if (history_f32) previous = V(history_f32[3 * p], history_f32[3 * p + 1], history_f32[3 * p + 2]);
      } else if (history_rgba8) {
        previous = V(lut[history_rgba8[4 * p]], lut[history_rgba8[4 * p + 1]], lut[history_rgba8[4 * p + 2]]);
      }
      uint32_t seg = 0;
      int32_t *hits = out_hits ? out_hits + p * R : NULL;
      if (hits) for (uint32_t i = 0; i < R; ++i) hits[i] = -2;
      v3 mixed = fs_main(&sc, x, y, previous, &seg, hits, g_ray_dump ? g_ray_dump + p * R * 6 : NULL);
      if (out_rgbf32) { out_rgbf32[3 * p] = mixed.x; out_rgbf32[3 * p + 1] = mixed.y; out_rgbf32[3 * p + 2] = mixed.z; }
      if (out_rgba8) {
        out_rgba8[4 * p] = srgb8_encode(mixed.x);
        out_rgba8[4 * p + 1] = srgb8_encode(mixed.y);
        out_rgba8[4 * p + 2] = srgb8_encode(mixed.z);
        out_rgba8[4 * p + 3] = 255; /* frag.hlsl:65: alpha 1. */
      }
      if (out_segments) out_segments[p] = seg;
    }
  }
  free(objs);
  (void)threads;
  return 0;
}

/* ---- single-function probes for the known-answer tests (tests/test_oracle_kat.py) */

/* One ray through ray_intersect: out = {distance, pos.xyz, normal.xyz}, returns object_index. */
RTO_EXPORT int rto_probe_intersect(const uint8_t *objects, uint32_t object_count, const float *origin,
                                   const float *direction, float *out7) {
  Scene sc;
  memset(&sc, 0, sizeof sc);
  Object *objs = (Object *)malloc(sizeof(Object) * (object_count ? object_count : 1));
  for (uint32_t i = 0; i < object_count; ++i) objs[i] = parse_object(objects + (size_t)i * OBJECT_STRIDE);
  sc.objects = objs;
  sc.object_count = object_count;
  Ray r;
  r.origin = V(origin[0], origin[1], origin[2]);
  r.direction = V(direction[0], direction[1], direction[2]);
  r.energy = V(1.0f, 1.0f, 1.0f);
  Hit h = ray_intersect(&sc, &r);
  out7[0] = h.distance;
  out7[1] = h.position.x; out7[2] = h.position.y; out7[3] = h.position.z;
  out7[4] = h.normal.x; out7[5] = h.normal.y; out7[6] = h.normal.z;
  free(objs);
  return h.object_index;
}

/* create_ray for normalised coordinate (u, v): out = direction.xyz */
RTO_EXPORT void rto_probe_create_ray(const uint8_t *config112, float u, float v, float *out3) {
  Scene sc;
  memset(&sc, 0, sizeof sc);
  sc.config = parse_config(config112);
  camera_setup(&sc);
  Ray r = create_ray(&sc, u, v);
  out3[0] = r.direction.x; out3[1] = r.direction.y; out3[2] = r.direction.z;
}

/* n calls of random() for rng_pixel = (px, py): out_values[n], out_seeds[n] (seed after each call) */
RTO_EXPORT void rto_probe_random(const float *random, uint32_t w, uint32_t h, float px, float py, uint32_t n,
                                 float *out_values, float *out_seeds) {
  Scene sc;
  memset(&sc, 0, sizeof sc);
  sc.random = random; sc.random_w = w; sc.random_h = h;
  Rng rng;
  random_init(&rng, px, py);
  for (uint32_t i = 0; i < n; ++i) {
    out_values[i] = rng_random(&sc, &rng);
    out_seeds[i] = rng.seed;
  }
}

/* texel (i, j) the HDRI lookup of direction d addresses */
RTO_EXPORT void rto_probe_hdri_texel(const float *d3, uint32_t w, uint32_t h, uint32_t *out_ij) {
  float u = fmaf(rto_atan2(d3[0], d3[2]), RTO_INV_TWO_PI, 0.5f);
  float v = fmaf(rto_asin(-d3[1]), RTO_INV_PI, 0.5f);
  out_ij[0] = texel_index(u, w);
  out_ij[1] = texel_index(v, h);
}

/* random_in_hemisphere(normal, roughness) with rng_pixel = (px, py): out = direction.xyz */
RTO_EXPORT void rto_probe_hemisphere(const float *random, uint32_t w, uint32_t h, float px, float py,
                                     const float *normal3, float roughness, float *out3) {
  Scene sc;
  memset(&sc, 0, sizeof sc);
  sc.random = random; sc.random_w = w; sc.random_h = h;
  Rng rng;
  random_init(&rng, px, py);
  v3 r = random_in_hemisphere(&sc, &rng, V(normal3[0], normal3[1], normal3[2]), roughness);
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

RTO_EXPORT uint8_t rto_srgb8_encode(float c) { return srgb8_encode(c); }
RTO_EXPORT float rto_srgb8_decode(uint32_t b) { return srgb8_decode(b); }

/* math probes (tests/test_oracle_math.py): op 0 sin, 1 cos, 2 tan, 3 asin, 4 log2, 5 exp2 */
RTO_EXPORT void rto_math_unary(int op, const float *x, float *y, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float s, c;
    switch (op) {
      case 0: rto_sincos(x[i], &s, &c); y[i] = s; break;
      case 1: rto_sincos(x[i], &s, &c); y[i] = c; break;
      case 2: y[i] = rto_tan(x[i]); break;
      case 3: y[i] = rto_asin(x[i]); break;
      case 4: y[i] = rto_log2(x[i]); break;
      default: y[i] = rto_exp2(x[i]); break;
    }
  }
}
/* op 0 atan2(a, b), 1 pow(a, b) */
RTO_EXPORT void rto_math_binary(int op, const float *a, const float *b, float *y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = op == 0 ? rto_atan2(a[i], b[i]) : rto_pow(a[i], b[i]);
}

RTO_EXPORT int rto_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

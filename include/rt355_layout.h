/* rt355_layout.h — wire formats that cross the rt355 C-ABI, as packed PODs.
 *
 * These are the byte layouts the reference's Rust serialisers emit for the WGSL/HLSL
 * bind group.  A caller keeps using its own writers
 * (Scene::as_bytes, FrameData::as_bytes) and hands the blobs over unchanged.
 *
 *   rt_wire_object   112 B  src/ray_tracer/objects.rs:41-57,125-157,202-214  (HLSL inputs.hlsl:1-23)
 *   rt_wire_light     48 B  src/ray_tracer/objects.rs:244-277                (HLSL inputs.hlsl:25-32; never read)
 *   rt_wire_config   112 B  src/ray_tracer/scene.rs:252-271                  (HLSL inputs.hlsl:34-50)
 *   rt_wire_frame     16 B  src/gpu/connection.rs:19,30-40                   (HLSL inputs.hlsl:52-55)
 * Little-endian, IEEE binary32; padding bytes are written as zero and ignored on read.
 */
#ifndef RT355_LAYOUT_H
#define RT355_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RT355_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT355_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct rt_wire_object {
  float colour[3];          /*   0 material.colour                  objects.rs:47 */
  uint32_t _pad0;           /*  12                                   objects.rs:48 */
  float emission[3];        /*  16 material.emission                objects.rs:49 */
  float emission_strength;  /*  28                                   objects.rs:50 */
  float metallic;           /*  32 (uploaded, unused by the shader) objects.rs:51 */
  float roughness;          /*  36                                   objects.rs:52 */
  uint32_t _pad1[2];        /*  40 implicit zero fill               bytes.rs:34   */
  uint32_t option;          /*  48 0 = sphere, 1 = plane            objects.rs:132,146 */
  uint32_t _pad2[3];        /*  52                                   objects.rs:133,147 */
  float center[3];          /*  64                                   objects.rs:134,148 */
  uint32_t _pad3;           /*  76                                   objects.rs:135,149 */
  float vec_data[3];        /*  80 plane normal (0 for a sphere)    objects.rs:150 */
  float f32_data;           /*  92 sphere radius / plane half size  objects.rs:136,151 */
  uint32_t _pad4[4];        /*  96 data[2] + tail                   inputs.hlsl:17 */
} rt_wire_object;

typedef struct rt_wire_light {
  uint32_t option;          /*   0 0 = direction, 1 = point */
  uint32_t _pad0[3];
  float intensity[3];       /*  16 */
  uint32_t _pad1;
  float vec_data[3];        /*  32 direction | position */
  uint32_t _pad2;
} rt_wire_light;

typedef struct rt_wire_config {
  float position[3];        /*   0 */
  uint32_t _pad0;
  float forward[3];         /*  16 rays leave along -forward (ray.hlsl:20) */
  uint32_t _pad1;
  float right[3];           /*  32 */
  uint32_t _pad2;
  float up[3];              /*  48 */
  uint32_t _pad3;
  float background_colour[3]; /* 64 (unused by the shader) */
  uint32_t _pad4;
  float ambient_light[3];   /*  80 (unused by the shader) */
  float fov;                /*  92 degrees */
  uint32_t reflection_limit;/*  96 */
  uint32_t width;           /* 100 */
  uint32_t height;          /* 104 */
  uint32_t _pad5;           /* 108 */
} rt_wire_config;

typedef struct rt_wire_frame {
  float jitter[2];          /*   0 in [-0.495, 0.495) (connection.rs:20,392-399) */
  uint32_t progressive_count; /* 8 */
  uint32_t _pad0;
} rt_wire_frame;

RT355_STATIC_ASSERT(sizeof(rt_wire_object) == 112, "Object stride is 112 B");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, emission) == 16, "emission @16");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, emission_strength) == 28, "emission_strength @28");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, roughness) == 36, "roughness @36");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, option) == 48, "geometry.option @48");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, center) == 64, "geometry.center @64");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, vec_data) == 80, "geometry.vec_data @80");
RT355_STATIC_ASSERT(offsetof(rt_wire_object, f32_data) == 92, "geometry.f32_data @92");
RT355_STATIC_ASSERT(sizeof(rt_wire_light) == 48, "Light stride is 48 B");
RT355_STATIC_ASSERT(sizeof(rt_wire_config) == 112, "Config is 112 B");
RT355_STATIC_ASSERT(offsetof(rt_wire_config, fov) == 92, "fov @92");
RT355_STATIC_ASSERT(offsetof(rt_wire_config, reflection_limit) == 96, "reflection_limit @96");
RT355_STATIC_ASSERT(offsetof(rt_wire_config, width) == 100, "width @100");
RT355_STATIC_ASSERT(offsetof(rt_wire_config, height) == 104, "height @104");
RT355_STATIC_ASSERT(sizeof(rt_wire_frame) == 16, "FrameData is 16 B");

enum {
  RT355_OBJECT_STRIDE = 112,
  RT355_LIGHT_STRIDE = 48,
  RT355_CONFIG_SIZE = 112,
  RT355_FRAME_SIZE = 16,
  RT355_REFERENCE_MAX_OBJECTS = 80, /* scene.rs:28 */
  RT355_REFERENCE_MAX_LIGHTS = 2    /* scene.rs:31 */
};

#endif /* RT355_LAYOUT_H */

/*
 * gvt_oracle.h -- CPU restatement of the GraviT engine-adapter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gravit_amd/ (the product) may
 * include, link, import or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the timed CPU baseline, never as a fallback.
 *
 * Parity status: PINNED on the reference's two golden images
 * (Test/CTESTtest/data/{simple,bunny}.ppm, tolerance CMakeLists.txt:666) in
 * smooth-normal mode, and on outputs of the reference's own Shade / Light /
 * Mesh::generateNormals / RayPacketIntersection / RandEngine code compiled
 * from /root/reference into oracle/_ref (see oracle/Makefile, ref_shim.cpp).
 * The triangle query itself lives in Embree >= 2.15 (2.x API, un-vendored and
 * un-pinned submodule third-party/embree, CMakeLists.txt:618): restated here
 * from its published Moeller-Trumbore intersector (see orc_tri_test).
 *
 * All layouts are the reference's binary layouts.
 */
#ifndef GVT_ORACLE_H
#define GVT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gvt::render::actor::Ray (actor/Ray.h:68-96): 80 B, 16-B aligned */
typedef struct {
  float origin[3];
  float t_min;
  float direction[3];
  float t_max;
  float color[3];
  float t;
  int32_t id;
  int32_t depth;
  float w;
  int32_t type; /* 0 PRIMARY, 1 SHADOW, 2 SECONDARY (Ray.h:50-55) */
  float pad[4];
} orc_ray;

/* gvt::render::data::primitives::Material (Material.h:59-90): 92 B */
typedef struct {
  int32_t type; /* 0 LAMBERT, 1 PHONG, 2 BLINN, 3 EMBREE_MATERIAL_METAL, 4 _VELVET, 5 _MATTE (Material.h:50-57) */
  float ka[3];
  float ks[3];
  float kd[3];
  float alpha;
  float eta[3];
  float k[3];
  float roughness;
  float horizonScatteringColor[3];
  float backScattering;
  float horizonScatteringFallOff;
} orc_material;

/* Tagged light POD (the CPU reference has virtual classes, scene/Light.h:46-104;
 * the POD precedent is adapter/optix/Light.cuh:44-112).  64 B. */
enum { ORC_LIGHT_POINT = 0, ORC_LIGHT_AREA = 1, ORC_LIGHT_AMBIENT = 2 };
typedef struct {
  int32_t type;
  float position[3];
  float color[3];
  float normal[3];
  float width, height;
  float pad[4];
} orc_light;

typedef struct {
  float t;
  int32_t prim; /* -1 = miss */
  float u, v;
} orc_hit;

typedef struct orc_mesh orc_mesh;

/* ---- mesh / acceleration structure (EmbreeMeshAdapter.cpp:125-162) ---- */
orc_mesh *orc_mesh_create(const float *verts, size_t nV, const int32_t *tris, size_t nT,
                          const float *vnormals /* nV*3 or NULL -> generateNormals */,
                          const float *vcolors /* nV*3 or NULL */, const orc_material *materials, size_t nMat,
                          const int32_t *face_mat /* nT or NULL; -1 = none */,
                          const orc_material *mesh_mat /* NULL -> default Material() */);
void orc_mesh_destroy(orc_mesh *);
const float *orc_mesh_normals(const orc_mesh *); /* nV*3 vertex normals in use */
void orc_mesh_bbox(const orc_mesh *, float lo[3], float hi[3]);

/* Mesh::generateNormals (Mesh.cpp:116-154) */
void orc_generate_normals(const float *verts, size_t nV, const int32_t *tris, size_t nT, float *normals_out);

/* closest hit of object-space rays against the mesh, t in (tnear, inf);
 * use_bvh=0 -> brute force over every triangle (same arithmetic). */
void orc_intersect(const orc_mesh *, const float *org /*n*3*/, const float *dir /*n*3*/, size_t n, float tnear,
                   int use_bvh, orc_hit *out);
void orc_occluded(const orc_mesh *, const float *org, const float *dir, size_t n, float tnear, int use_bvh,
                  int32_t *out /* 1 occluded */);

/* ---- the adapter boundary: Adapter::trace (Adapter.h:82-84, EmbreeMeshAdapter.cpp:436-660) ---- */
/* returns 0, or -1 if cap was too small (n_out then holds the needed count) */
int orc_trace(const orc_mesh *, orc_ray *rays /* in place, like the reference */, size_t begin, size_t end,
              orc_ray *rays_out, size_t cap, size_t *n_out, const float m[16], const float minv[16],
              const float normi[9], const orc_light *lights, size_t nLights, int normal_mode /*0 flat,1 smooth*/,
              uint32_t seed, int nthreads);

/* counts of rays pushed through the closest-hit and any-hit queries by the last orc_trace */
void orc_trace_counts(uint64_t *closest, uint64_t *anyhit);

/* ---- shading primitives, exposed for known-answer tests ---- */
int orc_shade(const orc_material *mat, const orc_ray *ray, const float N[3], const orc_light *light,
              const float lightPos[3], float color_out[3]); /* Material.cpp:90-139 */
void orc_light_contribution(const orc_light *, const float hit[3], const float samplePos[3], float out[3]);
float orc_rng(uint32_t *seed);                                   /* RandEngine.h:43-56 */
float orc_fastrand_lcg(uint32_t *seed, float mn, float mx);       /* RandEngine.h:78-81 */
/* CosWeightedRandomHemisphereDirection2 (EmbreeMeshAdapter.cpp:289-318): advances *seed by two draws */
void orc_cos_weighted_dir(const float n[3], uint32_t *seed, float out[3]);
/* include/gvt_math.h evaluated on the host, element-wise: kind 0 gvt_sinf(x), 1 gvt_cosf(x), 2 (float)gvt_acos(sqrt(1.0 - x)),
 * and the libm calls the reference makes in their place: 16 sinf, 17 cosf, 18 (float)acos(sqrt(1.0 - x)) */
void orc_math_probe(int kind, const float *in, size_t n, float *out);

/* ---- camera (gvtCamera.cpp:89-171, 233-312) ---- */
void orc_camera_generate(const float eye[3], const float focus[3], const float up[3], float fov, int width,
                         int height, int samples, int depth, float jitterWindowSize, orc_ray *rays_out);

/* ---- top-level instance acceleration (accel/BVH.cpp:77-216, BVH.h:61-135, RayPacket.h:83-211) ---- */
/* order_out[k] = instance id visited k-th by the reference's BVH traversal (DFS leaf order) */
void orc_toplevel_order(const float *inst_lo /*n*3*/, const float *inst_hi, size_t n, int32_t *order_out);
/* per ray: next instance (or -1) and entry distance, skipping instance `from` */
void orc_toplevel_intersect(const float *inst_lo, const float *inst_hi, const int32_t *order, size_t nInst,
                            const orc_ray *rays, size_t n, int from, int32_t *next_out, float *t_out);

/* ---- whole-frame schedulers (algorithm/ImageTracer.h:127-269, DomainTracer.h:185-496,
 *      TracerBase.h:325-414, IceTComposite.cpp:79-157) ---- */
typedef struct {
  const orc_mesh *const *meshes;   /* per instance */
  const float *m, *minv, *normi;   /* nInst*16, nInst*16, nInst*9 */
  const float *inst_lo, *inst_hi;  /* nInst*3 world AABBs (api.cpp:309-312) */
  size_t nInst;
  const orc_light *lights;
  size_t nLights;
  int normal_mode;
  int nthreads;
} orc_scene;

typedef struct {
  uint64_t rays_closest, rays_any, adapter_calls, rays_sent, rounds;
} orc_frame_stats;

/* Image scheduler, 1 rank.  fb = W*H*4 floats (RGBA), zeroed by the call. */
void orc_render_image(const orc_scene *, orc_ray *camera_rays /* consumed */, size_t nRays, int width, int height,
                      float *fb, orc_frame_stats *stats);
/* Domain scheduler simulated over P virtual ranks (instance i lives on rank owner[i]); the per-rank
 * framebuffers are summed and clamped (the build's composite) into fb. */
/* The build's image-identical shortcut of shuffleRays (gvt_oracle.c "known misses"; NOT reference behaviour, off by default): */
void orc_set_skip_known_misses(int on);
int orc_get_skip_known_misses(void);
/* shuffleRays' decision for n rays leaving instance `from` (in place: origins advanced, known-miss lists updated); next_out[i] = the
 * instance ray i goes on in, or -1 */
void orc_shuffle_step(const float *inst_lo, const float *inst_hi, const int32_t *order, size_t nInst, orc_ray *rays, size_t n, int from,
                      int32_t *next_out);
void orc_render_domain(const orc_scene *, const int32_t *owner, int P, const orc_ray *camera_rays, size_t nRays,
                       int width, int height, float *fb, orc_frame_stats *stats);

/* IceTComposite::write (IceTComposite.cpp:119-157): rows flipped, (uchar)(c*255); out = W*H*3 bytes */
void orc_fb_to_ppm_bytes(const float *fb, int width, int height, unsigned char *out);

#ifdef __cplusplus
}
#endif
#endif

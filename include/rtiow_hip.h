/*
 * rtiow_hip.h -- C ABI of the MI355X (gfx950) path-tracer core.
 *
 * The reference application has no plugin/FFI layer; its seam is the abstract
 * class `Raytracer` (R/src/Core/Raytracer.h:24-49: ctor(img, Scene&, H, W, spp,
 * maxDepth, buildUp), Run(), Cancel()) that `runRaytracer()` instantiates
 * (R/src/RaytracingApplication.cpp:260-269), and the per-frame compute dispatch
 * of the GPU toggle (R/src/RaytracingApplication.cpp:198-208).  Every entry point
 * below names the reference code it replaces.  R/ = Raytracing-In-A-Weekend/.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success
 * and a negative rt_status on failure (rt_last_error() gives the text); nothing
 * throws across the boundary; the library copies what it needs during the call and
 * never frees caller memory.  A scene handle is host-only and needs no GPU.  A
 * context owns one device and one stream; calls on one context are not re-entrant,
 * except rt_cancel(), which may be called from any thread.
 */
#ifndef RTIOW_HIP_H
#define RTIOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_scene rt_scene;     /* host-side scene under construction  */
typedef struct rt_context rt_context; /* one device + one stream + buffers   */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / handle / index            */
    RT_ERR_NO_DEVICE = -2, /* no usable HIP device (the path never falls back to the CPU) */
    RT_ERR_HIP = -3,       /* a HIP runtime call failed                */
    RT_ERR_STATE = -4,     /* call made in the wrong order             */
    RT_ERR_CANCELLED = -5, /* rt_cancel() interrupted the render       */
    RT_ERR_LIMIT = -6      /* size beyond a documented maximum         */
} rt_status;

/* Material kinds: R/src/Material/Material.h:17-170 */
enum {
    RT_MAT_LAMBERTIAN = 0, /* Material.h:17-40   */
    RT_MAT_METAL = 1,      /* Material.h:42-58   */
    RT_MAT_DIELECTRIC = 2, /* Material.h:60-96   */
    RT_MAT_LIGHT = 3,      /* Material.h:98-114  */
    RT_MAT_PBR = 4         /* Material.h:116-170 */
};

const char* rt_last_error(void);
const char* rt_version(void);

/* ------------------------------------------------------------------ scene (host) */

/* New empty scene: an empty world list (R/src/Core/Raytracer.h:17-22 `Scene`) plus a
 * default-seeded mt19937 that every later build step draws from in program order, as
 * the reference's process-global generator does (R/src/Core/RTWeekend.h:26-31). */
int rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);

/* Next n values of the scene's generator, exactly as randomFloat() returns them
 * (R/src/Core/RTWeekend.h:26-31).  Advances the stream. */
int rt_scene_rng_floats(rt_scene* s, int n, float* out);

/* Lambertian(a) / Metal(a, fuzz) / Dielectric(ir) / DiffuseLight(emit) constructors
 * (R/src/Material/Material.h:20,45,63,101).  Returns the material index (>= 0). */
int rt_scene_add_material(rt_scene* s, int kind, float r, float g, float b, float param);
/* Texture(path) after decoding: tightly packed RGB8, row 0 first (R/src/Material/Texture.h:7-29;
 * the 8-bit -> float map of stbi_loadf, R/vendor/stb_image/stb_image.h:1840-1861, is applied
 * by the library).  Returns the texture index. */
int rt_scene_add_texture_rgb8(rt_scene* s, int width, int height, const uint8_t* rgb);
/* The 256 floats the byte values 0..255 become when a texture is loaded: stbi_loadf's LDR -> float
 * map (R/vendor/stb_image/stb_image.h:1840-1861, called from R/src/Material/Texture.h:11), which
 * the kernels apply per channel when they sample.  Host only. */
int rt_texture_lut256(float* out256);
/* Texture(path) (R/src/Material/Texture.h:7-29): reads and decodes a JPEG (sequential or progressive) or a
 * PNG (interlaced or not) file with the library's own decoder -- whose bytes equal stbi_load()'s of the stb_image v2.27 the
 * reference vendors -- and keeps three channels per texel (grey replicated, alpha dropped; the
 * reference copies the first three floats of every texel whatever the file's channel count,
 * Texture.h:17-22, which is only meaningful for 3-channel files).  Returns the texture index. */
int rt_scene_add_texture_file(rt_scene* s, const char* path);
/* The decoder by itself: a JPEG / PNG file held in memory -> width, height, channel count as stored
 * (1, 2, 3 or 4) and, when `pixels` is not NULL, width * height * channels bytes, top row first.
 * Call once with pixels = NULL for the sizes.  Arithmetic-coded, lossless and 12-bit JPEG files are refused
 * (as stb_image refuses them); four-component files come out as three channels, as from stb_image. */
int rt_image_decode(const uint8_t* file_bytes, size_t n_bytes, int* width, int* height, int* channels, uint8_t* pixels,
                    size_t pixels_capacity);
/* PBRMaterial(diffuse) + setRoughnessTexture + setNormalTexture (Material.h:119,162-163);
 * pass -1 for an absent texture.  Returns the material index. */
int rt_scene_add_material_pbr(rt_scene* s, int tex_diffuse, int tex_roughness, int tex_normal);

/* world.add(make_shared<Sphere>(center, radius, material))  (R/src/Core/Hittable.h:226).
 * Returns the sphere id (creation order). */
int rt_scene_add_sphere(rt_scene* s, float cx, float cy, float cz, float radius, int material);
/* world.add(make_shared<Triangle>(v0, v1, v2, ...))  (R/src/Core/Hittable.h:50-56).
 * pos[9]; normal[9] (first normal all zero = geometric normal, Hittable.h:89-92);
 * uv[6] (uv[0] == -1 = use barycentrics, Hittable.h:97-107).  Returns the triangle id. */
int rt_scene_add_triangle(rt_scene* s, const float* pos9, const float* normal9, const float* uv6, int material);
/* world.add(make_shared<Box>(p0, p1, material, model))  (R/src/Core/Hittable.h:318-385):
 * twelve triangles in the reference's order.  model16 is column-major. */
int rt_scene_add_box(rt_scene* s, const float* p0, const float* p1, int material, const float* model16);
/* world.add(make_shared<Mesh>(model, file)) with the importer's output handed in as arrays
 * (R/src/Core/Mesh.cpp:7-37,106-208): per-face Triangle creation with the model-matrix
 * transform, mesh AABB, and BVHNode(triangles, 0, n, max_depth) built on the scene's
 * generator (the reference passes max_depth = 10).  normals / uvs may be NULL.
 * indices: 3 per triangle.  Returns the id of the mesh's first triangle. */
int rt_scene_add_mesh(rt_scene* s, const float* model16, int n_vertices, const float* positions, const float* normals,
                      const float* uvs, int n_triangles, const uint32_t* indices, int material, int max_depth);
/* Reads a glTF 2.0 file (.gltf with its buffers in files or "data:" URIs, or a .glb container) with one mesh
 * primitive and adds it as above
 * (the part of Assimp::Importer::ReadFile that R/src/Core/Mesh.cpp:10-12 relies on).
 * Texture images are not decoded here; bind them through `material`. */
int rt_scene_add_mesh_gltf(rt_scene* s, const float* model16, const char* gltf_path, int material, int max_depth);

/* world.add(make_shared<Mesh>(model, file)) for a glTF 2.0 file, textures included: the whole of
 * Mesh::Mesh (R/src/Core/Mesh.cpp:7-37) with the library's own glTF reader and image decoder in place of
 * the importer and stb_image (.gltf or .glb; buffers and images in files beside it, in "data:" URIs or in the
 * container's binary chunk).  Every triangle-list primitive of every mesh, in the order processNode
 * visits them (Mesh.cpp:39-52; node transforms are not applied there, nor here), becomes part of ONE
 * triangle list with one mesh box and one BVHNode(triangles, 0, n, max_depth).
 *   material < 0, per_primitive_materials = 0: the reference's rule (Mesh.cpp:56-104): one PBRMaterial
 *     for the whole Mesh from the first base-colour texture met; metallic-roughness and normal maps
 *     met from then on are attached to it (each image file is decoded once).
 *   material >= 0: that material for every triangle; no image is read.
 *   per_primitive_materials = 1 (beyond the reference, which shares one material per Mesh): every
 *     primitive gets its own material -- PBR from its own textures, or Lambertian(baseColorFactor).
 * Returns the id of the mesh's first triangle. */
int rt_scene_add_mesh_file(rt_scene* s, const float* model16, const char* gltf_path, int max_depth, int material,
                           int per_primitive_materials);

/* Ingestion of a Hittable tree the caller has ALREADY built -- e.g. a Scene whose world holds a
 * BVHNode or a Mesh -- without re-building it (no random number is drawn, the caller's boxes are used
 * as they are).  Everything added between a begin call and the matching rt_scene_end_group() becomes a
 * child, in order, of that group instead of the world; groups nest.
 *   rt_scene_begin_bvh_node(box): a BVHNode with its public `box` (R/src/AccelerationStructures/Bvh.h:
 *     96-99): add `left`, then `right` (once, when both point at the same object: a span-1 node tests
 *     its only child twice with the same outcome).  Its Hit() -- box test, left, then right against
 *     the closer of tMax and left's hit (Bvh.h:80-88) -- is a box test followed by a closest-hit scan.
 *   rt_scene_begin_mesh(box): a Mesh with its BoundingBox() (R/src/Core/Mesh.h:13-37): add the
 *     members of its `triangles` vector in order (after construction that is the one BVH root).
 *   rt_scene_begin_list(): a HittableList (R/src/Core/Hittable.h:272-316): add `objects` in order.
 * box = {min.x, min.y, min.z, max.x, max.y, max.z}.  Returns a non-negative group id. */
int rt_scene_begin_bvh_node(rt_scene* s, const float* box6);
int rt_scene_begin_mesh(rt_scene* s, const float* box6);
int rt_scene_begin_list(rt_scene* s);
int rt_scene_end_group(rt_scene* s);

/* Instancing.  NO COUNTERPART IN THE REFERENCE, which has no instances: Mesh::processMesh bakes the model matrix
 * into every vertex (R/src/Core/Mesh.cpp:106-208), so N placements of a mesh are N triangle lists and N trees.
 * Opt-in, like normal mapping, and defined by the CPU oracle's Instance class: everything added between
 * rt_scene_begin_prototype() and rt_scene_end_prototype() (which returns the prototype's number) forms an object
 * in its own space that is NOT in the world; rt_scene_add_instance() puts a placement of it where an object
 * would go (the world or an open group; rt_scene_wrap_bvh() builds the reference's BVHNode over placements
 * like over any other object).  A placement's Hit(): its world-space box (the prototype's box corners under the
 * model matrix), then the ray taken into object space -- direction NOT normalised, so hit distances are the same
 * numbers in both spaces --, the prototype's own Hit(), and the normal brought back through
 * mat3(transpose(inverse(model))) and normalised; the hit point is taken from the world ray.
 * One level only (no instance inside a prototype).  Scenes with instances are rendered along the reference-order
 * entries (they have no search structure) by a build of the persistent loop that walks the two levels, and not
 * together with normal mapping.
 * model16: column-major.  Returns the instance's number. */
int rt_scene_begin_prototype(rt_scene* s);
int rt_scene_end_prototype(rt_scene* s);
int rt_scene_add_instance(rt_scene* s, int prototype, const float* model16);

/* world = randomScene()  (R/src/RaytracingApplication.cpp:359-405), drawn in g++ argument order. */
int rt_scene_random_scene(rt_scene* s);
/* world = cornellBox()   (R/src/RaytracingApplication.cpp:407-478). */
int rt_scene_cornell_box(rt_scene* s);
/* world = HittableList(make_shared<BVHNode>(world))  (R/src/AccelerationStructures/Bvh.h:17-78). */
int rt_scene_wrap_bvh(rt_scene* s);

/* Camera(lookfrom, lookat, vup, vfov, aspectRatio, aperture, focusDist)  (R/src/Core/Camera.h:7-25). */
int rt_scene_set_camera(rt_scene* s, const float* lookfrom3, const float* lookat3, const float* vup3, float vfov,
                        float aspect_ratio, float aperture, float focus_dist);
/* The eight camera members in the order the reference uploads them as uniforms
 * (R/src/RaytracingApplication.cpp:140-147): origin, lowerLeftCorner, horizontal, vertical,
 * u, v, w (3 floats each), lensRadius. */
int rt_scene_get_camera(const rt_scene* s, float* out22);
/* The same eight members handed over as they are, in that order: what `Raytracer`'s constructor copies out
 * of Scene& (`mCamera = scene.camera`, R/src/Core/Raytracer.h:28; the members are R/src/Core/Camera.h:35-38)
 * and what the GPU toggle uploads as uniforms (R/src/RaytracingApplication.cpp:140-147).  A caller that
 * holds a constructed Camera has only these -- the constructor's arguments are gone, and recovering them
 * would go through atan/tan and not round-trip -- so the renderer takes them bit for bit:
 * rt_scene_get_camera() returns exactly what was set.  No value is checked or normalised. */
int rt_scene_set_camera_members(rt_scene* s, const float* in22);
int rt_scene_set_background(rt_scene* s, float r, float g, float b); /* Scene::background */
/* Normal mapping for PBR materials that carry a normal map.  Off (default): as the reference renders,
 * whose PBRMaterial::scatter forms the mapped normal and then overwrites it with rec.normal
 * ("Normalmapping doesn't work properly yet", R/src/Material/Material.h:123-140).  On: the mapped
 * normal normalize(TBN * (2 texel - 1)) is used for the reflection, the diffuse direction and the
 * facing test, with the per-triangle tangent frame of R/src/Core/Hittable.h:110-125 (each uv
 * difference paired with the edge to the same vertex) and the shading normal; the normal map's
 * texels are read as byte / 255, not through the gamma map meant for colours.  Changes the image,
 * so it has no counterpart output in the reference; takes effect at the next rt_upload_scene(). */
int rt_scene_set_normal_mapping(rt_scene* s, int on);

/* ------------------------------------------------- flattened scene (host, for inspection) */

typedef struct rt_flat_info {
    int n_nodes;      /* threaded BVH entries, reference pre-order              */
    int n_prim_refs;  /* leaf references in reference visit order               */
    int n_spheres;
    int n_triangles;
    int n_materials;
    int n_textures;
    int max_leaf;     /* longest run of primitives under one entry              */
    int fits_lds;     /* 1 if the traversal data fits the LDS-resident kernel   */
} rt_flat_info;

/* Flattens the world (the BVH flatten step): packed node array in the reference's
 * pre-order (BVHNode::Hit's left-then-right order, Bvh.h:80-88), leaf references in visit
 * order.  Idempotent; called implicitly by rt_upload_scene(). */
int rt_scene_flatten(rt_scene* s, rt_flat_info* info);
/* Copies of the flattened arrays.  nodes: 8 x 32-bit words per entry
 * {min.x, min.y, min.z, skip | flags, max.x, max.y, max.z, first_ref};
 * flag bit 30 of word 3 = entry has no box test (a HittableList, Hittable.h:281-296).
 * prim_refs: bit 31 = last of its leaf, bit 30 = triangle, low bits = id.  Any pointer may be NULL. */
int rt_scene_get_flat(const rt_scene* s, uint32_t* nodes8, uint32_t* prim_refs, float* spheres4, int32_t* sphere_mat,
                      float* tri_pos9, float* tri_normal9, float* tri_uv6, int32_t* tri_mat, int32_t* mat_kind,
                      float* mat_rgbp4);

/* The structure the kernels search first (no counterpart in the reference: a tighter tree over
 * the same primitives whose answer is checked against the reference's boxes, see rt_render).
 * entries: 8 x 32-bit words {min.x, max.x, min.y, max.y, min.z, max.z, skip, first_ref}, in walk
 * order (an entry whose box passes is followed by entry + 1, one whose box fails by `skip`);
 * first_ref = 0xFFFFFFFF for an entry without a leaf run.  refs: as prim_refs.  Sizes come back in
 * n_entries / n_refs; either array may be NULL (call once for the sizes, once for the data). */
int rt_scene_get_search_structure(const rt_scene* s, int* n_entries, uint32_t* entries8, int* n_refs, uint32_t* refs);
/* Scenes too large for LDS (and up to 2 M entries) carry the same entries eight times, each copy
 * with the children of every entry ordered near-to-far for one sign pattern of the ray direction
 * (copy k: bit a of k set = component a negative); a ray walks the copy for its signs.  A copy has
 * `stride` entries (the last one a box no ray passes); links index the whole array and
 * 8 * stride means "walk over"; leaf runs are the refs of rt_scene_get_search_structure.
 * stride comes back 0 when the scene has no such copies; entries8 may be NULL. */
int rt_scene_get_search_structure_octants(const rt_scene* s, int* stride, uint32_t* entries8);

/* ------------------------------------------------------------------ device */

int rt_device_count(void);
/* Opens device `device` (hipSetDevice) and creates the context's stream. */
int rt_create(int device, rt_context** out);
void rt_destroy(rt_context* ctx);

/* Makes the context issue all later work on `hip_stream` (a hipStream_t the caller owns, on the
 * context's device) instead of its own stream, so that a caller's collectives and these kernels
 * are ordered on one stream without host synchronisation.  NULL restores the context's stream. */
int rt_set_stream(rt_context* ctx, void* hip_stream);

/* Diagnostics: with on = 1 every ray walks the flattened node array in the reference's order
 * (BVHNode::Hit, R/src/AccelerationStructures/Bvh.h:80-88).  With on = 0 (default) rays first
 * search a tighter tree over the same primitives and fall back to that walk only when the
 * guard test of the winner fails; both settings return identical hits.  Applies to later calls. */
int rt_set_reference_traversal(rt_context* ctx, int on);

/* Where later rt_upload_scene() calls build the search structure: on = 1 ON THE DEVICE (csrc/device_bvh.hip:
 * Morton-sorted primitives clustered bottom-up by the surface area of their unions, the top of the tree split
 * top-down, then the host builder's rule for leaving out inner entries), on = 0 on the host (binned surface-area
 * heuristic, fast_bvh.cpp), on = -1 (the default) on the device for scenes of one million primitives or more that
 * do not carry a host-built structure yet.  The device build takes milliseconds where the host takes seconds
 * (10 M triangles: 23 ms against 2.7 s) and its tree costs 6 to 8 % more per frame.  No counterpart in the reference; the reference-order tree
 * (BVHNode's constructor, R/src/AccelerationStructures/Bvh.h:18-78, sequential by its use of one random stream
 * and std::sort) is built on the host either way, and images do not depend on this switch (the guard test against
 * the reference's boxes decides every hit).
 * rt_last_build_ms(): device time of the last such build.  rt_read_search_structure(): the entries and
 * leaf references the context's kernels walk (layout of rt_scene_get_search_structure), read back from
 * the device; sizes first with NULL arrays. */
int rt_set_device_build(rt_context* ctx, int on);
int rt_last_build_ms(rt_context* ctx, float* ms);
int rt_read_search_structure(rt_context* ctx, int* n_entries, uint32_t* entries8, int* n_refs, uint32_t* refs);

/* Copies the flattened scene, camera and background to the device (replaces what
 * `Raytracer`'s constructor captures from Scene&, Raytracer.h:27-31).  Returns RT_ERR_LIMIT if the search
 * structure has 2^27 entries or more (the kernels address its entries by 32-bit byte offsets). */
int rt_upload_scene(rt_context* ctx, rt_scene* s);

/* Image size and the share of it this context renders.  The image is cut into 8x8-pixel
 * tiles; tile (tx, ty) belongs to rank (tx + ty) mod world.  Allocates and zeroes the
 * accumulation buffer (mOrigColorData, Raytracer.h:28-31).  rank 0 / world 1 = whole image. */
int rt_set_target(rt_context* ctx, int width, int height, int rank, int world);
/* Same, but the accumulation buffer is caller-owned device memory of rt_accum_bytes() bytes
 * (256-byte aligned), e.g. a tensor that a collective library will read.  Not zeroed. */
int rt_set_target_external(rt_context* ctx, int width, int height, int rank, int world, void* device_accum);
size_t rt_accum_bytes(int width, int height, int world);
/* Gives the accumulation buffer up (the end of mOrigColorData's life, R/src/Core/Raytracer.h:38): waits for
 * the stream, frees a buffer the context allocated, forgets a caller-owned one.  Afterwards rt_render(),
 * rt_clear() and the read calls fail with RT_ERR_STATE until the next rt_set_target().  For callers that
 * are about to free an external buffer. */
int rt_release_target(rt_context* ctx);

/* Zeroes the accumulation buffer (the memset of Raytracer.h:30). */
int rt_clear(rt_context* ctx);

/* Adds samples [first_sample, first_sample + n_samples) of every owned pixel to the
 * accumulation buffer: the pixel loops of RaytracerNormal::Run / RaytracerMT::writeLine
 * (R/src/Core/Raytracer.cpp:46-132) with rayColor (Raytracer.cpp:3-20) for each sample.
 * Random numbers are the reference shader's stateless hash keyed (x, y, sample, counter)
 * (R/assets/shaders/raytracer.comp:35-54).  Every sample's radiance passes through a device
 * colour buffer (12 bytes per pixel sample of the launch, allocated on first use and grown on
 * demand; calls that would need more than RT_COLORS_MB megabytes, default 8192, run as several
 * launches) and is added per pixel in sample order, so the sums have the bits of the reference's
 * sequential `pixelColor +=`.  Launches of 48 M samples and more of a scene whose tables fit LDS run as two
 * kernels -- the camera rays of a tile walked together, then the loop over the surviving paths -- and keep 48
 * more bytes per pixel sample for the path records between them (RT_RECORDS_MB, default 25600, cuts the launch;
 * RT_TWO_PASS=0 / 1: never / always).  Asynchronous on the context's stream unless a buffer must grow. */
int rt_render(rt_context* ctx, int first_sample, int n_samples, int max_depth);
/* rt_render() for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same work,
 * same bits as rt_render(), but the per-sample colour buffer and the queue's counters live in a fixed
 * array inside the library (64 MiB + 4 KiB per GPU in use, held until the process ends), so the
 * call allocates and frees nothing; a sample range that needs more colour space than that runs as
 * several launches.  Fails with RT_ERR_LIMIT when the context's share of the frame has more than
 * rt_noalloc_max_pixels() pixel slots (5,592,405 = 64 MiB / 12 bytes: one sample of every slot must fit).  Every call on
 * one GPU uses that GPU's array: calls that overlap in time on one GPU (other contexts, other
 * streams) race and must be serialised by the caller. */
int rt_render_noalloc(rt_context* ctx, int first_sample, int n_samples, int max_depth);
size_t rt_noalloc_max_pixels(void);
/* Blocks until the stream is idle (the join in runRaytracer(), RaytracingApplication.cpp:269). */
int rt_sync(rt_context* ctx);
/* Raytracer::Cancel() (Raytracer.h:59-62,77-80; the reference polls its flag once per scanline,
 * Raytracer.cpp:54,78,148,176).  Makes a running rt_run() stop: the flag lives in pinned host memory that
 * the running kernel's waves read every time they fetch 64 samples of work, so a pass stops within the
 * time its longest paths in flight take, not at the next pass boundary.  A Cancel() that arrives before
 * rt_run() starts stops that run; the flag is cleared when rt_run() returns and by rt_set_target().
 * Thread-safe; the one call that may be made while another call on the context is running. */
int rt_cancel(rt_context* ctx);

/* Complete progressive render: passes of `samples_per_pass` until `samples_per_pixel` are
 * done or rt_cancel() is called; after each pass `on_pass(done, user)` may be invoked
 * (NULL = no callback).  Stands in for Raytracer::Run() (Raytracer.cpp:46-95,134-194).
 * Cancelled: returns RT_ERR_CANCELLED.  A kernel launch that the flag cut short adds nothing to the
 * accumulation buffer, so every pixel always holds the same number of samples; if that is more than
 * the last callback reported (a pass that ran as several launches), the callback is invoked once more
 * with that number before rt_run() returns. */
typedef void (*rt_pass_callback)(int samples_done, void* user);
int rt_run(rt_context* ctx, int samples_per_pixel, int samples_per_pass, int max_depth, rt_pass_callback on_pass,
           void* user);

/* Row-major W*H*4 float sums (alpha = 1), row 0 = bottom of the image: the RGBA32F image
 * the GPU toggle binds (RaytracingApplication.cpp:132-133).  Pixels of other ranks are 0
 * unless rt_resolve_gathered() was used.  Blocking. */
int rt_read_accum_rgba32f(rt_context* ctx, float* dst);
/* writeColor()'s bytes (Raytracer.cpp:22-44): 256 * clamp(sqrt(sum / samples_per_pixel), 0, 0.999),
 * A = 255, row 0 = bottom: the buffer glTexImage2D uploads (RaytracingApplication.cpp:194). */
int rt_read_rgba8(rt_context* ctx, int samples_per_pixel, uint8_t* dst);
/* Writes a W*H*4 RGBA8 buffer laid out as rt_read_rgba8() returns it (row 0 = bottom) as a plain
 * PPM ("P3", 255, top row first, alpha dropped).  Host only; needs no device.  The reference hands
 * its bytes to glTexImage2D (RaytracingApplication.cpp:194) and has no file writer; this is the
 * image.ppm of the book the reference follows, for headless use. */
int rt_write_ppm(const char* path, int width, int height, const uint8_t* rgba8);
/* The same buffer as an 8-bit RGB PNG (top row first, alpha dropped; the pixel data goes into stored, i.e.
 * uncompressed, deflate blocks, which every PNG reader accepts).  Host only.  Like rt_write_ppm it stands where
 * the reference uploads its bytes to a texture (RaytracingApplication.cpp:194) and writes no file. */
int rt_write_png(const char* path, int width, int height, const uint8_t* rgba8);

/* Device-side resolve of `world` rank payloads laid end to end (the output of an all-gather
 * of every rank's accumulation buffer) into a row-major RGBA32F device image. */
int rt_resolve_gathered(rt_context* ctx, const void* device_payloads, int world, void* device_rgba32f);

/* The same resolve straight to writeColor()'s bytes (Raytracer.cpp:22-44): a row-major W*H RGBA8
 * device image, row 0 = bottom, ready for one device-to-host copy into the buffer glTexImage2D uploads. */
int rt_resolve_gathered_rgba8(rt_context* ctx, const void* device_payloads, int world, int samples_per_pixel,
                              void* device_rgba8);

/* ------------------------------------------------------------------ several GPUs, one process
 *
 * No counterpart in the reference (a single-device application; its render thread is one host thread,
 * R/src/RaytracingApplication.cpp:244-276): a group drives n devices of one node from ONE host process
 * and owns the RCCL communicator between them (loaded at run time; RT_ERR_STATE when librccl cannot
 * be loaded).  The scene is replicated; device i renders the 8x8 tiles with (tx + ty) mod n == i; a read
 * call all-gathers the n payloads over xGMI (one ncclAllGather per frame), resolves them on device 0
 * and copies the frame to the host.  The frame has the same bits for every n.  devices = NULL means
 * devices 0 .. n-1.  Errors of these calls: rt_group_last_error(). */
typedef struct rt_group rt_group;
int rt_group_create(int n_devices, const int* devices, rt_group** out);
void rt_group_destroy(rt_group* g);
int rt_group_size(const rt_group* g);
rt_context* rt_group_context(rt_group* g, int i);      /* the context of device i (e.g. for rt_count) */
int rt_group_upload_scene(rt_group* g, rt_scene* s);
int rt_group_set_target(rt_group* g, int width, int height);
int rt_group_clear(rt_group* g);
int rt_group_render(rt_group* g, int first_sample, int n_samples, int max_depth); /* asynchronous, all devices */
int rt_group_read_rgba8(rt_group* g, int samples_per_pixel, uint8_t* dst);        /* gather + resolve + copy; blocking */
int rt_group_read_accum_rgba32f(rt_group* g, float* dst);
const char* rt_group_last_error(void);

/* Test hook: closest hit of n rays (origin xyz, direction xyz) with tMin = 0.001, as
 * mWorld.Hit(r, 0.001f, infinity, rec) (Raytracer.cpp:10).  prim_kind: -1 miss, 0 sphere,
 * 1 triangle.  Optional outputs may be NULL.  Host pointers; blocking. */
int rt_trace(rt_context* ctx, int n, const float* rays6, int32_t* prim_kind, int32_t* prim_id, float* t, float* normal3,
             int32_t* front_face, float* uv2);

/* Test hook: n independent Material::scatter() calls (R/src/Material/Material.h:22-160) with the
 * hit record handed in: material index, incoming ray, hit point, shading normal, frontFace,
 * (u, v), and the RNG position key4 = {pixel x, pixel y, sample, counter before the call}.
 * Outputs: attenuation, scattered ray, alive flag (scatter()'s return value) and the RNG
 * counter after the call.  Host pointers; blocking. */
int rt_scatter_test(rt_context* ctx, int n, const int32_t* material, const float* ray_in6, const float* p3,
                    const float* normal3, const int32_t* front_face, const float* uv2, const uint32_t* key4,
                    float* atten3, float* ray_out6, int32_t* alive, uint32_t* counter_out);

/* Test hook: Dielectric::reflectance(cosine, refIdx) (R/src/Material/Material.h:90-95, Schlick's
 * approximation with pow() in double) for n pairs, as the kernels evaluate it.  Host pointers; blocking. */
int rt_reflectance_test(rt_context* ctx, int n, const float* cosine, const float* ref_idx, float* out);

/* Path statistics of one counting pass over the owned pixels (same paths as rt_render):
 * samples, rays, node tests, sphere tests, triangle tests, hits, scatters by material kind
 * (lambertian, metal, dielectric, light, pbr), misses, depth cut-offs.  Node and primitive
 * test counts are those of the traversal actually run (see rt_set_reference_traversal).  Blocking. */
typedef struct rt_counters {
    uint64_t samples, rays, node_tests, sphere_tests, tri_tests, hits;
    uint64_t scatter_lambertian, scatter_metal, scatter_dielectric, scatter_light, scatter_pbr;
    uint64_t misses, depth_cutoffs;
    uint64_t fallbacks; /* rays re-traced in the reference's order (guard test failed, NaN, zero direction component) */
} rt_counters;
int rt_count(rt_context* ctx, int first_sample, int n_samples, int max_depth, rt_counters* out);

/* Diagnostics of the wavefront scheduler, filled by the last rt_count(): 7 pairs {wave-level
 * executions, lanes that took part}: the times each operation was voted in (search-structure box
 * test, primitive test, shade, new path), the wave-uniform walk steps of a two-pass launch's first
 * pass, then the box-test steps inside the loop's votes, then the votes for the reference-order box
 * test.  No reference counterpart. */
int rt_sched_stats(rt_context* ctx, uint64_t* out14);

/* Milliseconds the kernels of the last rt_render() took on the device (HIP events on the
 * context's stream around the render and ordered-sum kernels).  Blocks until they have finished. */
int rt_last_render_ms(rt_context* ctx, float* ms);
/* The context's hipStream_t, for callers that order their own work after rt_render(). */
void* rt_stream(rt_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTIOW_HIP_H */

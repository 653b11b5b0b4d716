/* A caller of include/rtiow_hip.h written in C and built with the system C compiler: what a binding
 * on the reference side would compile.  (The Python tests reach the same library through ctypes with
 * signatures typed by hand; this file is checked by the compiler against the header itself.)
 *
 *   capi_driver layout                 prints sizeof / offsetof of the ABI's structs; needs no GPU
 *   capi_driver render W H SPP OUT     cover scene of R/src/RaytracingApplication.cpp:359-405 wrapped in a BVH,
 *                                      the book's camera, SPP samples at depth 50; writes the W*H*4 RGBA8
 *                                      bytes (rt_read_rgba8) to OUT and prints the path counters of sample 0
 * R/ = Raytracing-In-A-Weekend/ in the reference tree.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtiow_hip.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ < 0) {                                                       \
            fprintf(stderr, "%s: %d: %s\n", #call, rc_, rt_last_error());    \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static int layout(void) {
    printf("rt_counters %zu", sizeof(rt_counters));
    printf(" %zu %zu %zu %zu %zu %zu %zu", offsetof(rt_counters, samples), offsetof(rt_counters, rays),
           offsetof(rt_counters, node_tests), offsetof(rt_counters, sphere_tests), offsetof(rt_counters, tri_tests),
           offsetof(rt_counters, hits), offsetof(rt_counters, scatter_lambertian));
    printf(" %zu %zu %zu %zu %zu %zu %zu\n", offsetof(rt_counters, scatter_metal), offsetof(rt_counters, scatter_dielectric),
           offsetof(rt_counters, scatter_light), offsetof(rt_counters, scatter_pbr), offsetof(rt_counters, misses),
           offsetof(rt_counters, depth_cutoffs), offsetof(rt_counters, fallbacks));
    printf("rt_flat_info %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_flat_info), offsetof(rt_flat_info, n_nodes),
           offsetof(rt_flat_info, n_prim_refs), offsetof(rt_flat_info, n_spheres), offsetof(rt_flat_info, n_triangles),
           offsetof(rt_flat_info, n_materials), offsetof(rt_flat_info, n_textures), offsetof(rt_flat_info, max_leaf),
           offsetof(rt_flat_info, fits_lds));
    printf("status %d %d %d %d %d %d %d\n", RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_STATE,
           RT_ERR_CANCELLED, RT_ERR_LIMIT);
    return 0;
}

static void on_pass(int done, void* user) { *(int*)user = done; }

static int render(int w, int h, int spp, const char* out_path) {
    rt_scene* scene = NULL;
    rt_context* ctx = NULL;
    const float lookfrom[3] = {13.0f, 2.0f, 3.0f}, lookat[3] = {0.0f, 0.0f, 0.0f}, vup[3] = {0.0f, 1.0f, 0.0f};
    float cam[22], cam2[22];
    rt_flat_info info;
    rt_counters c;
    int passes_seen = 0;
    size_t bytes = (size_t)w * (size_t)h * 4;
    uint8_t* img = (uint8_t*)malloc(bytes);
    FILE* f;
    if (!img) return 1;
    CHECK(rt_scene_create(&scene));
    CHECK(rt_scene_random_scene(scene));
    CHECK(rt_scene_wrap_bvh(scene));
    /* aspectRatio = imageWidth / imageHeight is an integer division in the reference (RaytracingApplication.cpp:251) */
    CHECK(rt_scene_set_camera(scene, lookfrom, lookat, vup, 20.0f, (float)(w / h), 0.1f, 10.0f));
    /* hand the derived members over a second time, as an adapter that only holds a Camera would */
    CHECK(rt_scene_get_camera(scene, cam));
    CHECK(rt_scene_set_camera_members(scene, cam));
    CHECK(rt_scene_get_camera(scene, cam2));
    if (memcmp(cam, cam2, sizeof cam) != 0) {
        fprintf(stderr, "camera members did not round-trip\n");
        return 1;
    }
    CHECK(rt_scene_set_background(scene, 0.7f, 0.8f, 1.0f));
    CHECK(rt_scene_flatten(scene, &info));
    CHECK(rt_create(0, &ctx));
    CHECK(rt_upload_scene(ctx, scene));
    CHECK(rt_set_target(ctx, w, h, 0, 1));
    CHECK(rt_run(ctx, spp, 1, 50, on_pass, &passes_seen));
    CHECK(rt_read_rgba8(ctx, spp, img));
    CHECK(rt_count(ctx, 0, 1, 50, &c));
    f = fopen(out_path, "wb");
    if (!f || fwrite(img, 1, bytes, f) != bytes || fclose(f) != 0) {
        fprintf(stderr, "cannot write %s\n", out_path);
        return 1;
    }
    printf("spheres %d nodes %d passes %d samples %llu rays %llu node_tests %llu sphere_tests %llu hits %llu misses %llu\n",
           info.n_spheres, info.n_nodes, passes_seen, (unsigned long long)c.samples, (unsigned long long)c.rays,
           (unsigned long long)c.node_tests, (unsigned long long)c.sphere_tests, (unsigned long long)c.hits,
           (unsigned long long)c.misses);
    rt_destroy(ctx);
    rt_scene_destroy(scene);
    free(img);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 2 && strcmp(argv[1], "layout") == 0) return layout();
    if (argc == 6 && strcmp(argv[1], "render") == 0) return render(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argv[5]);
    fprintf(stderr, "usage: %s layout | render W H SPP OUT\n", argv[0]);
    return 2;
}

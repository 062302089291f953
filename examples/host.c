/* A C host over include/rt_amd.h, the way the reference's main.cpp would bind the library (reference
 * src/main.cpp:358 builds the tree, :498-500 renders rectangles, :157-167 converts to bytes):
 *
 *     host <scene.rtscene> <width> <height> [out.ppm]
 *
 * loads a baked scene, renders one frame with the reference's default options on GPU 0 and prints a checksum of the
 * 8-bit image (FNV-1a 64) and the ray counts.  Built by examples/Makefile with plain gcc; links librt_amd.so only.
 *
 * .rtscene layout (little endian; written by raytracer_amd/scene.py): 8 bytes "RTSCN01\0", uint32 counts {meshes,
 * materials, lights}, then records that ARE the C-ABI structs: one rt_camera_desc (64 bytes), the rt_material_desc
 * array (64 bytes each), the rt_light_desc array (32 bytes each), and per mesh a 32-byte name, int32 material, int32
 * n_tris, n_tris*9 floats of corner positions and as many of corner normals. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_amd.h"

static int fail(const char* what) { fprintf(stderr, "host: %s: %s\n", what, rt_last_error()); return 1; }

static int read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: host scene.rtscene width height [out.ppm]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char magic[8]; uint32_t cnt[3];
    if (!read_exact(f, magic, 8) || memcmp(magic, "RTSCN01", 8) != 0 || !read_exact(f, cnt, 12)) { fprintf(stderr, "not an .rtscene file\n"); return 2; }
    const int n_meshes = (int)cnt[0], n_mats = (int)cnt[1], n_lights = (int)cnt[2];
    rt_camera_desc cam;
    rt_material_desc* mats = calloc((size_t)n_mats, sizeof *mats);
    rt_light_desc* lights = calloc((size_t)(n_lights > 0 ? n_lights : 1), sizeof *lights);
    if (sizeof cam != 64 || sizeof *mats != 64 || sizeof *lights != 32) { fprintf(stderr, "struct layout differs from the file format\n"); return 2; }
    if (!read_exact(f, &cam, sizeof cam) || !read_exact(f, mats, (size_t)n_mats * sizeof *mats) ||
        !read_exact(f, lights, (size_t)n_lights * sizeof *lights)) { fprintf(stderr, "short file\n"); return 2; }
    rt_mesh_desc* meshes = calloc((size_t)n_meshes, sizeof *meshes);
    for (int i = 0; i < n_meshes; ++i) {
        if (!read_exact(f, meshes[i].name, 32) || !read_exact(f, &meshes[i].material, 4) || !read_exact(f, &meshes[i].n_tris, 4)) return 2;
        const size_t nb = (size_t)meshes[i].n_tris * 9 * sizeof(float);
        float* pos = malloc(nb); float* nrm = malloc(nb);
        if (!pos || !nrm || !read_exact(f, pos, nb) || !read_exact(f, nrm, nb)) return 2;
        meshes[i].pos = pos; meshes[i].nrm = nrm;
    }
    fclose(f);

    if (rt_init(0) != RT_OK) return fail("rt_init");
    rt_scene* scene = NULL;
    if (rt_scene_create(meshes, n_meshes, mats, n_mats, lights, n_lights, &cam, &scene) != RT_OK) return fail("rt_scene_create");
    int64_t shape[4];
    rt_scene_info(scene, shape);

    rt_options opt; memset(&opt, 0, sizeof opt);
    opt.width = atoi(argv[2]); opt.height = atoi(argv[3]);
    opt.hfov = 54.43f; opt.spp = 2; opt.spr = 4; opt.max_bounce = 3; opt.max_refrac_bounce = 3;   /* src/main.cpp:74-75,103-111 */
    opt.eps = 1e-4f; opt.ray_eps = 0.1f; opt.jitter_mode = RT_JITTER_OFF; opt.seed = 1; opt.exact_boxes = 1;
    const size_t n_px = (size_t)opt.width * opt.height;
    float* rgb = malloc(n_px * 3 * sizeof(float));
    uint8_t* img = malloc(n_px * 3);
    rt_stats st;
    /* the reference renders the frame as disjoint rectangles (src/main.cpp:474-507); two halves here */
    const int half = opt.height / 2;
    if (rt_render_rect(scene, &opt, 0, 0, opt.width, half, rgb, &st, NULL) != RT_OK) return fail("rt_render_rect");
    uint64_t closest = st.rays_closest, any = st.rays_any;
    if (rt_render_rect(scene, &opt, 0, half, opt.width, opt.height, rgb + (size_t)half * opt.width * 3, &st, NULL) != RT_OK) return fail("rt_render_rect");
    closest += st.rays_closest; any += st.rays_any;
    if (rt_resolve_u8(rgb, opt.width, opt.height, img, NULL) != RT_OK) return fail("rt_resolve_u8");
    /* a bad call is reported, not fatal */
    if (rt_render_rect(scene, &opt, 0, 0, opt.width + 1, 1, rgb, NULL, NULL) != RT_ERR_INVALID) { fprintf(stderr, "host: a rectangle outside the image was accepted\n"); return 1; }

    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n_px * 3; ++i) { h ^= img[i]; h *= 1099511628211ull; }
    printf("triangles %lld nodes %lld depth %lld\n", (long long)shape[0], (long long)shape[1], (long long)shape[3]);
    printf("rays_closest %llu rays_any %llu\n", (unsigned long long)closest, (unsigned long long)any);
    printf("fnv64 %016llx\n", (unsigned long long)h);
    if (argc > 4) {
        FILE* o = fopen(argv[4], "wb");
        if (!o) { perror(argv[4]); return 2; }
        fprintf(o, "P6\n%d %d\n255\n", opt.width, opt.height);
        fwrite(img, 1, n_px * 3, o);
        fclose(o);
    }
    rt_scene_destroy(scene);
    return 0;
}

// The reference's main() without its window (src/main.cpp:700-777): scene folder and option file in, render_result.jpg
// out, the report lines of renderFrame (src/main.cpp:445-461,521-523) on stdout and in logs.txt.  A C++ host over
// include/rt_amd.h only - scene loading, option parsing, tree build, rendering, 8-bit conversion and the JPEG all go
// through the C-ABI of librt_amd.so; compiled with g++, no HIP headers.
//
//   raytracer_host <scene folder | scene.fbx | scene.rtscene> [options.txt] [-o render_result.jpg] [--index N]
//                  [--seed N] [--jitter off|hash] [--exact] [--blocks] [--gpus N] [--log logs.txt]
//
// A folder is searched for scene files in name order (retrieve_files, src/main.cpp:218-237); with several the
// reference asks which one on stdin (:281-289), here --index picks (default 0).  --blocks renders display block by
// display block (-mpbs pixels square, src/main.cpp:474-507) instead of in one call; --gpus N fans the frame out over N
// GPUs from this one process (one scene handle per GPU, rt_render_frame_multi) the way the reference fans it out over
// threads.  --exact asks for the reference's own box arithmetic and leaves (bit-equal to it, slower).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../include/rt_amd.h"

static FILE* g_log = nullptr;

// src/main.cpp:125-138: every line to stdout and to the log file
static void logprintf(const char* fmt, ...) {
    va_list a, b;
    va_start(a, fmt); va_copy(b, a);
    vprintf(fmt, a);
    if (g_log) vfprintf(g_log, fmt, b);
    va_end(a); va_end(b);
}

static bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && strcasecmp(s.c_str() + s.size() - n, suffix) == 0;
}

static int die(const char* what, const char* detail) { fprintf(stderr, "raytracer_host: %s: %s\n", what, detail); return 1; }

int main(int argc, char** argv) {
    std::string scene_arg, option_file, out_path = "render_result.jpg", log_path;
    int index = 0, gpus = 1, exact = 0, jitter = RT_JITTER_HASH, blocks = 0;
    unsigned seed = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); } return argv[++i]; };
        if (a == "-o") out_path = next("-o");
        else if (a == "--index") index = atoi(next("--index"));
        else if (a == "--seed") seed = (unsigned)strtoul(next("--seed"), nullptr, 10);
        else if (a == "--jitter") jitter = strcmp(next("--jitter"), "off") == 0 ? RT_JITTER_OFF : RT_JITTER_HASH;
        else if (a == "--exact") exact = 1;
        else if (a == "--blocks") blocks = 1;
        else if (a == "--gpus") gpus = atoi(next("--gpus"));
        else if (a == "--log") log_path = next("--log");
        else if (scene_arg.empty()) scene_arg = a;
        else if (option_file.empty()) option_file = a;
        else { fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
    }
    if (scene_arg.empty() || gpus < 1) {
        fprintf(stderr, "usage: raytracer_host <scene folder | file> [options.txt] [-o render_result.jpg] [--index N] [--seed N] "
                        "[--jitter off|hash] [--exact] [--blocks] [--gpus N] [--log logs.txt]\n");
        return 2;
    }
    if (!log_path.empty()) g_log = fopen(log_path.c_str(), "a");

    // ---- options (src/main.cpp:718-736)
    rt_options opt;
    rt_options_default(&opt);
    int display_block = 100;                                             // prog_disp_span, src/main.cpp:115
    if (!option_file.empty()) {
        FILE* f = fopen(option_file.c_str(), "r");
        if (!f) return die("cannot open the option file", option_file.c_str());
        char line[512];
        while (fgets(line, sizeof(line), f)) rt_options_parse_line(&opt, line, &display_block);
        fclose(f);
    }
    opt.seed = seed; opt.jitter_mode = jitter; opt.exact_boxes = exact;

    // ---- scene file (src/main.cpp:256-293)
    std::string scene_path = scene_arg;
    struct stat info;
    if (stat(scene_arg.c_str(), &info) != 0) return die("cannot access", scene_arg.c_str());
    if (S_ISDIR(info.st_mode)) {
        std::vector<std::string> files;
        if (DIR* d = opendir(scene_arg.c_str())) {
            while (dirent* e = readdir(d)) { const std::string n = e->d_name; if (ends_with(n, ".fbx") || ends_with(n, ".rtscene")) files.push_back(n); }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
        if (files.empty()) return die("no scene file in", scene_arg.c_str());
        logprintf("Detected Scene Files:\n");
        for (size_t k = 0; k < files.size(); ++k) logprintf("%d. %s\n", (int)k, files[k].c_str());
        if (index < 0 || (size_t)index >= files.size()) return die("--index beyond the files of", scene_arg.c_str());
        scene_path = scene_arg + "/" + files[(size_t)index];
    }
    logprintf("Loading scene: %s\n", scene_path.c_str());
    rt_host_scene* host = nullptr;
    if (rt_host_scene_load(scene_path.c_str(), &host) != RT_OK) return die("scene import failed", rt_host_last_error());
    const rt_mesh_desc* meshes; const rt_material_desc* mats; const rt_light_desc* lights; const rt_camera_desc* cam;
    int n_meshes, n_mats, n_lights;
    rt_host_scene_desc(host, &meshes, &n_meshes, &mats, &n_mats, &lights, &n_lights, &cam);
    logprintf("\nDetected Meshes:\n\n");
    static const char* kind_line[4] = {"Can't find a proper material defined for this material, using default", "Associated this mesh with lambert material",
                                       "Associated this mesh with phong material", "Associated this mesh with simple refractive material"};
    long long n_tris = 0;
    for (int i = 0; i < n_meshes; ++i) {
        logprintf("%s: material %d\n%s (%d triangles)\n", kind_line[mats[meshes[i].material].kind & 3], meshes[i].material, meshes[i].name, meshes[i].n_tris);
        n_tris += meshes[i].n_tris;
    }
    logprintf("\nDetected Lights:\n\n");
    for (int i = 0; i < n_lights; ++i)
        logprintf("%s: %.2f %.2f %.2f; Color: %.2f %.2f %.2f;\n", lights[i].kind == RT_LIGHT_DIRECTIONAL ? "Direction (toward the light)" : "Position",
                  lights[i].vec[0], lights[i].vec[1], lights[i].vec[2], lights[i].color[0], lights[i].color[1], lights[i].color[2]);

    // ---- one scene handle per GPU (src/main.cpp:358 builds the tree once; here once per device)
    logprintf("\nConstructing AABB Tree for hit detection...");
    std::vector<rt_scene*> scenes((size_t)gpus, nullptr);
    for (int g = 0; g < gpus; ++g) {
        if (rt_init(g) != RT_OK) return die("rt_init", rt_last_error());
        if (rt_scene_create(meshes, n_meshes, mats, n_mats, lights, n_lights, cam, &scenes[(size_t)g]) != RT_OK) return die("rt_scene_create", rt_last_error());
    }
    logprintf("done\n");

    // ---- renderFrame (src/main.cpp:440-526)
    const time_t now = time(nullptr);
    logprintf("\n[Rendering Report]\nTime: %sRendering starts.\n", ctime(&now));
    logprintf("Compute: %d GPU%s, %lld triangles\n", gpus, gpus > 1 ? "s" : "", n_tris);
    logprintf("Samples per pixel: %d\n", opt.spp * opt.spp);
    logprintf("Samples per ray: %d\n", opt.spr + 1);
    logprintf("Epsilon: %f\n", opt.eps);
    logprintf("Ray Jitter Epsilon: %f\n", opt.ray_eps);
    logprintf("Max bounces: %d\n", opt.max_bounce);
    logprintf("Max refractive bounces: %d\n", opt.max_refrac_bounce);
    logprintf("Field of view: %.2f\n", opt.hfov);
    logprintf("Resolution: %d X %d\n", opt.width, opt.height);
    const size_t n_px = (size_t)opt.width * opt.height;
    std::vector<float> rgb(n_px * 3);
    std::vector<uint8_t> img(n_px * 3);
    unsigned long long rays = 0, ref_calls = 0;
    logprintf("Drawing... ");
    const auto t0 = std::chrono::high_resolution_clock::now();
    rt_stats st;
    if (gpus > 1) {
        if (rt_render_frame_multi(scenes.data(), gpus, &opt, 8, rgb.data(), &st) != RT_OK) return die("rt_render_frame_multi", rt_last_error());
        rays = st.rays_closest + st.rays_any; ref_calls = st.ref_get_hit_calls;
    } else if (blocks) {
        // the reference's progressive order: blocks of -mpbs pixels, rows of blocks from the bottom (src/main.cpp:474-507)
        const int span = std::max(1, display_block);
        std::vector<float> part;
        for (int j = 0; j < opt.height; j += span)
            for (int i = 0; i < opt.width; i += span) {
                const int x1 = std::min(i + span, opt.width), y1 = std::min(j + span, opt.height);
                part.resize((size_t)(x1 - i) * (y1 - j) * 3);
                if (rt_render_rect(scenes[0], &opt, i, j, x1, y1, part.data(), &st, nullptr) != RT_OK) return die("rt_render_rect", rt_last_error());
                for (int y = j; y < y1; ++y) memcpy(&rgb[((size_t)y * opt.width + i) * 3], &part[(size_t)(y - j) * (x1 - i) * 3], (size_t)(x1 - i) * 12);
                rays += st.rays_closest + st.rays_any; ref_calls += st.ref_get_hit_calls;
            }
    } else {
        if (rt_render_rect(scenes[0], &opt, 0, 0, opt.width, opt.height, rgb.data(), &st, nullptr) != RT_OK) return die("rt_render_rect", rt_last_error());
        rays = st.rays_closest + st.rays_any; ref_calls = st.ref_get_hit_calls;
    }
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    logprintf("done\n");
    logprintf("Rendering finished with total duration of %f seconds.\n", secs);
    logprintf("Rays traced: %llu (the reference's loop would call get_hit %llu times), %.1f Mrays/s\n", rays, ref_calls, rays / secs / 1e6);

    // ---- save_jpg (src/main.cpp:148-180)
    if (rt_resolve_u8(rgb.data(), opt.width, opt.height, img.data(), nullptr) != RT_OK) return die("rt_resolve_u8", rt_last_error());
    if (ends_with(out_path, ".ppm")) {
        FILE* o = fopen(out_path.c_str(), "wb");
        if (!o) return die("cannot write", out_path.c_str());
        fprintf(o, "P6\n%d %d\n255\n", opt.width, opt.height);
        fwrite(img.data(), 1, img.size(), o);
        fclose(o);
    } else if (rt_save_jpeg(out_path.c_str(), img.data(), opt.width, opt.height, 90, "Render Result") != RT_OK) return die("cannot write", out_path.c_str());
    logprintf("Saved %s\n", out_path.c_str());
    for (rt_scene* s : scenes) rt_scene_destroy(s);
    rt_host_scene_free(host);
    if (g_log) fclose(g_log);
    return 0;
}

// main.cpp -- `volren`: the offline (--render) driver of the reference's src/main.cpp on the HIP renderer.
//
// Same command line, same order semantics (src/main.cpp:311-347 for -w/-h, :360-435 for everything else: arguments are
// processed left to right, paths load immediately, later flags override what a loader set), same per-frame loop
// (:524-557): reset -> sppx samples -> tonemap -> "<stem>_%06d.png" in the current directory.
// Not provided: the interactive window/GUI, Python scripts (.py arguments: `python -m volren_amd.run_script` runs them) and the tinycolormap presets.
// Addition: --gpus N [--devices a,b,...] renders every frame on N devices (sharded.h: scene replicated, 16x16 tiles dealt diagonally, one
// grouped ncclAllGather per frame; a device named more than once = logical shards of one GPU, exchanged by device-to-device copies).
//
//   volren data/smoke.brick data/table_mountain_2_puresky_1k.hdr -w 1024 -h 1024 --render --spp 4096 --bounces 128 \
//          --albedo 0.8 --phase 0.3 --density 100 --env_strength 3 --env_rot 270 --exposure 3 --gamma 2.0 --cam_fov 40
#include <sys/wait.h>

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <filesystem>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "imageio.h"
#include "renderer.h"
#include "sharded.h"

namespace fs = std::filesystem;
using namespace vr;

static std::shared_ptr<RendererHIP> renderer;
static std::string out_filename = "output.png";

// A path on the command line is loaded the moment it is seen (src/main.cpp:37-81, 428-432): ".hdr" replaces the environment, ".txt" binds a
// transfer function and hides the environment, anything else is a volume file or a folder of animation frames -- which also resets the density
// scale, fits the volume into the unit cube and commits it.  Every kind restarts the accumulation; a loader that throws is reported on stderr
// with the reference's wording and the render goes on with what was loaded before (the reference's catch blocks).
struct PathKind {
    const char* ext;       // file extension, or nullptr for "everything else"
    const char* what;      // the noun of the reference's message: "Unable to load <what> from <path>: ..."
    void (*load)(const std::string& path);
};
static const PathKind kPathKinds[] = {
    { ".hdr", "envmap", [](const std::string& path) { renderer->environment = std::make_shared<Environment>(path); } },
    { ".txt", "transferfunc", [](const std::string& path) {
          renderer->transferfunc = std::make_shared<TransferFunction>(path);
          renderer->show_environment = false;
      } },
    { nullptr, "volume", [](const std::string& path) {
          std::cout << "load volume: " << path << std::endl;
          renderer->volume = fs::is_directory(path) ? Volume::load_folder(path) : std::make_shared<Volume>(path);
          renderer->density_scale = 1.f;
          renderer->scale_and_move_to_unit_cube();
          renderer->commit();
      } },
};
static void handle_path(const std::string& path) {
    const std::string ext = fs::path(path).extension().string();
    if (ext == ".py") return;                               // main() hands the whole command line to python -m volren_amd.run_script before anything else
    for (const PathKind& kind : kPathKinds) {
        if (kind.ext && ext != kind.ext) continue;
        try {
            kind.load(path);
            renderer->sample = 0;
        } catch (const std::exception& e) {
            std::cerr << "Unable to load " << kind.what << " from " << path << ": " << e.what() << std::endl;
        }
        return;
    }
}

struct Args {
    int argc; char** argv; int i;
    bool has(int n) const { return i + n < argc; }
    std::string next() { if (i + 1 >= argc) throw std::runtime_error(std::string("missing value after ") + argv[i]); return argv[++i]; }
    float nextf() { return std::stof(next()); }
    int nexti() { return std::stoi(next()); }
};

static void parse_cmd(int argc, char** argv) {
    Args a{ argc, argv, 0 };
    for (a.i = 1; a.i < argc; ++a.i) {
        const std::string arg = argv[a.i];
        if (arg == "--render") {
        } else if (arg == "-w" || arg == "-h" || arg == "--title" || arg == "--major" || arg == "--minor" || arg == "--swap" || arg == "--font" || arg == "--fontsize") {
            a.next();                                           // consumed by the context set-up pass
        } else if (arg == "--no-resize" || arg == "--hidden" || arg == "--no-decoration" || arg == "--floating" || arg == "--maximised" || arg == "---debug") {
        } else if (arg == "--output") out_filename = a.next();
        else if (arg == "--samples" || arg == "--spp" || arg == "--sppx") renderer->sppx = a.nexti();
        else if (arg == "--bounces") renderer->bounces = a.nexti();
        else if (arg == "--albedo") renderer->albedo = vec3(a.nextf());
        else if (arg == "--density") renderer->density_scale = a.nextf();       // SETS (after load_volume multiplied it)
        else if (arg == "--emission") renderer->emission_scale = a.nextf();
        else if (arg == "--phase") renderer->phase = a.nextf();
        else if (arg == "--env_strength") renderer->environment->strength = a.nextf();
        else if (arg == "--env_rot") renderer->environment->transform = rotation_axis(a.nextf(), 1);
        else if (arg == "--env_hide") renderer->show_environment = false;
        else if (arg == "--fau") {
            renderer->transferfunc = std::make_shared<TransferFunction>(std::vector<vec4>({ vec4(0, 0, 0, 0), vec4(4 / 255.f, 49 / 255.f, 106 / 255.f, 0.33f),
                                                                                          vec4(38 / 255.f, 97 / 255.f, 65 / 255.f, 0.66f), vec4(151 / 255.f, 27 / 255.f, 47 / 255.f, 1.f) }));
        } else if (arg == "--turbo" || arg == "--viridis") {
            std::cerr << arg << ": tinycolormap presets are not part of this build" << std::endl;
        // src/main.cpp:397-402: the value is only consumed when a transfer function exists; without one it stays behind as an argument of its own
        // (not a flag, not a file: ignored)
        } else if (arg == "--tf_left") { if (renderer->transferfunc) renderer->transferfunc->window_left = a.nextf(); }
        else if (arg == "--tf_width") { if (renderer->transferfunc) renderer->transferfunc->window_width = a.nextf(); }
        else if (arg == "--cam_pos") { renderer->camera.pos.x = a.nextf(); renderer->camera.pos.y = a.nextf(); renderer->camera.pos.z = a.nextf(); }
        else if (arg == "--cam_dir") { renderer->camera.dir.x = a.nextf(); renderer->camera.dir.y = a.nextf(); renderer->camera.dir.z = a.nextf(); }
        else if (arg == "--cam_fov") renderer->camera.fov_degree = a.nextf();
        else if (arg == "--exposure") renderer->tonemap_exposure = a.nextf();
        else if (arg == "--gamma") renderer->tonemap_gamma = a.nextf();
        else if (arg == "--vol_rot_x" || arg == "--vol_rot_y" || arg == "--vol_rot_z") {
            // glm::mat3(glm::rotate(mat4(volume->transform), angle, axis)): the translation is dropped (reference quirk, main.cpp:417-422)
            const int axis = arg.back() - 'x';
            const mat4 r = from3(rotation_axis(a.nextf(), axis));
            renderer->volume->transform = from3(upper3(renderer->volume->transform * r));
        } else if (arg == "--vol_crop_min") { renderer->vol_clip_min.x = a.nextf(); renderer->vol_clip_min.y = a.nextf(); renderer->vol_clip_min.z = a.nextf(); }
        else if (arg == "--vol_crop_max") { renderer->vol_clip_max.x = a.nextf(); renderer->vol_clip_max.y = a.nextf(); renderer->vol_clip_max.z = a.nextf(); }
        else if (arg == "--seed") renderer->seed = a.nexti();                        // addition
        else if (arg == "--device" || arg == "--gpus" || arg == "--devices") a.next();  // consumed earlier
        else if (fs::is_regular_file(arg) || fs::is_directory(arg)) handle_path(arg);
    }
}

static std::vector<int> parse_int_list(const std::string& s) {
    std::vector<int> v;
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t c = s.find(',', pos);
        const std::string tok = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        if (!tok.empty()) v.push_back(std::stoi(tok));
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return v;
}

// `volren script.py --render -w W -h H` (src/main.cpp:83-91 embeds CPython and evaluates the file): this build starts an ordinary interpreter on
// volren_amd.run_script, which makes `import volpy` resolve to the HIP-backed module and gives `volpy.Renderer()` the -w / -h resolution.  Done before
// this process has touched the GPU; the interpreter is a child, its exit status is ours.
static int run_python_script(int argc, char** argv) {
    std::error_code ec;
    const fs::path exe = fs::read_symlink("/proc/self/exe", ec);
    const std::string root = ec ? std::string(".") : exe.parent_path().parent_path().string();      // <root>/volren_amd/volren
    std::string cmd = "PYTHONPATH='" + root + "'${PYTHONPATH:+:$PYTHONPATH} python3 -m volren_amd.run_script";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], q = "'";
        for (char c : a) { if (c == '\'') q += "'\\''"; else q += c; }
        cmd += " " + q + "'";
    }
    const int rc = std::system(cmd.c_str());
    return rc == -1 ? 1 : (WIFEXITED(rc) ? WEXITSTATUS(rc) : 1);
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg.size() > 3 && arg.compare(arg.size() - 3, 3, ".py") == 0 && fs::is_regular_file(arg)) return run_python_script(argc, argv);
    }
    int width = 1280, height = 720, device = 0, gpus = 0;        // cppgl ContextParameters defaults (unverified): always pass -w/-h
    std::vector<int> devices;
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string arg = argv[i];
            if (arg == "-w" && i + 1 < argc) width = std::stoi(argv[++i]);
            else if (arg == "-h" && i + 1 < argc) height = std::stoi(argv[++i]);
            else if (arg == "--device" && i + 1 < argc) device = std::stoi(argv[++i]);
            else if (arg == "--gpus" && i + 1 < argc) gpus = std::stoi(argv[++i]);
            else if (arg == "--devices" && i + 1 < argc) devices = parse_int_list(argv[++i]);
        }
        // one renderer per part; without --gpus / --devices: one part on --device, the reference's single-context loop
        if (devices.empty()) {
            if (gpus <= 1) devices = { device };
            else for (int d = 0; d < gpus; ++d) devices.push_back(d);
        } else if (gpus > 0 && (size_t)gpus != devices.size()) throw std::runtime_error("--gpus and --devices disagree");
        const bool sharded = devices.size() > 1 || gpus > 0;
        std::vector<std::shared_ptr<RendererHIP>> parts;
        for (size_t k = 0; k < devices.size(); ++k) {
            // the scene is replicated by running the command line once per part (paths load onto the part's device)
            VR_HIP(hipSetDevice(devices[k]));
            renderer = std::make_shared<RendererHIP>();
            renderer->resolution = { width, height };
            renderer->init();
            parse_cmd(argc, argv);
            if (renderer->volume->grids.empty()) {
                // debug box of the reference (main.cpp:465-474): a 1x1x4 dense grid in front of the camera
                const float values[4] = { 1.f, 2.5f, 5.f, 10.f };
                auto box = std::make_shared<DenseGrid>(1, 1, 4, values);
                const vec3 d = renderer->camera.dir;
                box->transform = scale_then_translate(1.f, vec3(2.f * d.x + 0.f, 2.f * d.y - 0.5f, 2.f * d.z - 2.f));
                renderer->volume = std::make_shared<Volume>(box);
                renderer->commit();
            }
            renderer->reset();
            parts.push_back(renderer);
        }
        renderer = parts[0];                            // holds the whole frame after the gather
        std::unique_ptr<ShardedRenderer> shards;
        if (sharded) {
            std::vector<RendererHIP*> raw;
            for (auto& p : parts) raw.push_back(p.get());
            shards = std::make_unique<ShardedRenderer>(raw, devices);
            std::cout << "rendering on " << devices.size() << " part(s), devices";
            for (int d : devices) std::cout << " " << d;
            std::cout << ", tile exchange: " << shards->transport() << std::endl;
        }
        std::cout << "rendering..." << std::endl;
        for (size_t i = 0; i < renderer->volume->n_grid_frames(); ++i) {
            for (auto& p : parts) { p->reset(); p->volume->grid_frame_counter = i; }
            const auto t0 = std::chrono::steady_clock::now();
            if (shards) {
                shards->render(renderer->sppx);
                shards->synchronize();
            } else {
                renderer->render(renderer->sppx);          // == while (sample < sppx) trace();
                renderer->synchronize();
                if (renderer->watchdog_status()) throw std::runtime_error("path-tracing kernel watchdog tripped");
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << renderer->sample << " / " << renderer->sppx << "  (" << sec << " s, "
                      << (double)width * height * renderer->sppx / sec / 1e6 << " Msamples/s)" << std::endl;
            VR_HIP(hipSetDevice(devices[0]));
            renderer->tonemapping = true;               // the offline loop always tonemaps (main.cpp:540-550)
            renderer->draw();
            std::vector<float> fb((size_t)width * height * 4);
            renderer->download_display(fb.data());
            std::vector<uint8_t> rgba;
            framebuffer_to_rgba8(fb.data(), width, height, rgba);
            char num[16];
            snprintf(num, sizeof num, "%06zu", i);
            const std::string out_fn = fs::path(out_filename).stem().string() + "_" + num + ".png";   // directory of --output is dropped (reference quirk)
            save_png_rgba8(out_fn, rgba.data(), width, height);
            std::cout << out_fn << " written." << std::endl;
        }
        shards.reset();                                 // before the parts it points at
    } catch (std::exception& e) {
        std::cerr << "volren: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

/* glref.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Runs the reference's own GLSL compute shaders (shader/pathtracer_brick*.glsl, common.glsl, env_setup.glsl, tonemap.glsl,
 * read from /root/reference at fixture-generation time, never copied) on Mesa's software rasteriser (llvmpipe) in the
 * GPU-less build container, so that the CPU oracle can be pinned against outputs of the reference's kernels themselves
 * (tests/golden/make_golden_glsl.py -> tests/golden/glsl_*.npz).  There is no X server, EGL or OSMesa in the image: the
 * OpenGL 4.5 core context is created directly on the DRI swrast driver interface (GL/internal/dri_interface.h).
 *
 * The host side of the reference (src/renderer.cpp, src/environment.cpp) is NOT built -- it needs cppgl, voldata, GLFW,
 * imgui, none of which is vendored -- so this file restates, call by call, the GL object set-up those files perform
 * for this path (texture formats and filters: renderer.cpp:159-218; uniforms: renderer.cpp:88-138; importance map:
 * environment.cpp:11-37).  It is a thin C layer; the scene logic lives in oracle/glref/binding.py.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

static char g_err[8192];
const char* glref_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ---------------------------------------------------------------- context */
static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) {}
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { memset(data, 0, (size_t)w * h * 4); }
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) {}
static void get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) {}
static const __DRIswrastLoaderExtension swrast_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = get_drawable_info, .putImage = put_image, .getImage = get_image,
    .putImage2 = put_image2, .getImage2 = get_image2,
};
static const __DRIextension* loader_ext[] = { &swrast_loader.base, NULL };

#define GL_FUNCS(X) \
    X(PFNGLGETSTRINGPROC, glGetString) X(PFNGLGETERRORPROC, glGetError) X(PFNGLFINISHPROC, glFinish) \
    X(PFNGLCREATESHADERPROC, glCreateShader) X(PFNGLSHADERSOURCEPROC, glShaderSource) X(PFNGLCOMPILESHADERPROC, glCompileShader) \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv) X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) X(PFNGLCREATEPROGRAMPROC, glCreateProgram) \
    X(PFNGLATTACHSHADERPROC, glAttachShader) X(PFNGLLINKPROGRAMPROC, glLinkProgram) X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) X(PFNGLUSEPROGRAMPROC, glUseProgram) X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) \
    X(PFNGLUNIFORM1IPROC, glUniform1i) X(PFNGLUNIFORM1UIPROC, glUniform1ui) X(PFNGLUNIFORM1FPROC, glUniform1f) X(PFNGLUNIFORM2FPROC, glUniform2f) \
    X(PFNGLUNIFORM3FPROC, glUniform3f) X(PFNGLUNIFORM2IPROC, glUniform2i) X(PFNGLUNIFORMMATRIX3FVPROC, glUniformMatrix3fv) \
    X(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv) X(PFNGLGENTEXTURESPROC, glGenTextures) X(PFNGLBINDTEXTUREPROC, glBindTexture) \
    X(PFNGLTEXIMAGE3DPROC, glTexImage3D) X(PFNGLTEXIMAGE2DPROC, glTexImage2D) X(PFNGLTEXPARAMETERIPROC, glTexParameteri) \
    X(PFNGLACTIVETEXTUREPROC, glActiveTexture) X(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture) X(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute) \
    X(PFNGLMEMORYBARRIERPROC, glMemoryBarrier) X(PFNGLGETTEXIMAGEPROC, glGetTexImage) X(PFNGLGENERATEMIPMAPPROC, glGenerateMipmap) \
    X(PFNGLGENBUFFERSPROC, glGenBuffers) X(PFNGLBINDBUFFERPROC, glBindBuffer) X(PFNGLBUFFERDATAPROC, glBufferData) \
    X(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) X(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData) X(PFNGLPIXELSTOREIPROC, glPixelStorei) \
    X(PFNGLGETTEXLEVELPARAMETERIVPROC, glGetTexLevelParameteriv)
#define DECL(T, N) static T p_##N;
GL_FUNCS(DECL)

static int g_ready = 0;
int glref_init(void) {
    if (g_ready) return 0;
    void* h = dlopen("swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) FAIL("dlopen swrast_dri.so: %s", dlerror());
    const __DRIextension** (*get_ext)(void) = (const __DRIextension** (*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get_ext) FAIL("swrast_dri.so has no __driDriverGetExtensions_swrast");
    const __DRIextension** ext = get_ext();
    const __DRIcoreExtension* core = NULL; const __DRIswrastExtension* sw = NULL;
    for (int i = 0; ext[i]; ++i) {
        if (!strcmp(ext[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)ext[i];
        if (!strcmp(ext[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension*)ext[i];
    }
    if (!core || !sw || sw->base.version < 4) FAIL("DRI core / swrast (v4) extension missing");
    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = sw->createNewScreen2(0, loader_ext, ext, &configs, NULL);
    if (!screen || !configs || !configs[0]) FAIL("createNewScreen2 failed");
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
    unsigned err = 0;
    __DRIcontext* ctx = sw->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) FAIL("createContextAttribs(4.5 core) failed: %u", err);
    __DRIdrawable* draw = sw->createNewDrawable(screen, configs[0], NULL);
    if (!draw || !core->bindContext(ctx, draw, draw)) FAIL("bindContext failed");
    void* (*gpa)(const char*) = (void* (*)(const char*))dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!gpa) FAIL("no _glapi_get_proc_address");
#define LOAD(T, N) p_##N = (T)gpa(#N); if (!p_##N) FAIL("GL function %s missing", #N);
    GL_FUNCS(LOAD)
    g_ready = 1;
    return 0;
}
const char* glref_info(void) {
    static char buf[256];
    if (!g_ready) return "";
    snprintf(buf, sizeof buf, "%s | %s", (const char*)p_glGetString(GL_VERSION), (const char*)p_glGetString(GL_RENDERER));
    return buf;
}
static int check(const char* what) {
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) FAIL("GL error 0x%x in %s", e, what);
    return 0;
}

/* ---------------------------------------------------------------- programs */
int glref_program(const char* source) {          /* one compute shader; source already has its #includes expanded */
    GLuint sh = p_glCreateShader(GL_COMPUTE_SHADER);
    p_glShaderSource(sh, 1, &source, NULL);
    p_glCompileShader(sh);
    GLint ok = 0;
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[6000]; p_glGetShaderInfoLog(sh, sizeof log, NULL, log); FAIL("compile: %s", log); }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, sh);
    p_glLinkProgram(prog);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { char log[6000]; p_glGetProgramInfoLog(prog, sizeof log, NULL, log); FAIL("link: %s", log); }
    return (int)prog;
}
int glref_use(int prog) { p_glUseProgram((GLuint)prog); return check("glUseProgram"); }
/* uniforms by name; kind: 0 int, 1 uint, 2 float, 3 vec2, 4 vec3, 5 ivec2, 6 mat3, 7 mat4.  A name the linker removed is not an error (returns 1). */
int glref_uniform(int prog, const char* name, int kind, const void* v) {
    GLint loc = p_glGetUniformLocation((GLuint)prog, name);
    if (loc < 0) return 1;
    const float* f = (const float*)v; const int32_t* i = (const int32_t*)v;
    switch (kind) {
        case 0: p_glUniform1i(loc, i[0]); break;
        case 1: p_glUniform1ui(loc, (GLuint)i[0]); break;
        case 2: p_glUniform1f(loc, f[0]); break;
        case 3: p_glUniform2f(loc, f[0], f[1]); break;
        case 4: p_glUniform3f(loc, f[0], f[1], f[2]); break;
        case 5: p_glUniform2i(loc, i[0], i[1]); break;
        case 6: p_glUniformMatrix3fv(loc, 1, GL_FALSE, f); break;
        case 7: p_glUniformMatrix4fv(loc, 1, GL_FALSE, f); break;
        default: FAIL("bad uniform kind");
    }
    return check(name);
}
int glref_sampler(int prog, const char* name, int unit, int tex, int is3d) {      /* cppgl Shader::uniform(name, texture, unit) */
    GLint loc = p_glGetUniformLocation((GLuint)prog, name);
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(is3d ? GL_TEXTURE_3D : GL_TEXTURE_2D, (GLuint)tex);
    if (loc < 0) return 1;
    p_glUniform1i(loc, unit);
    return check(name);
}

/* ---------------------------------------------------------------- textures */
/* renderer.cpp:159-218.  kind 0: indirection GL_RGB10_A2UI, 1: range GL_RG16F (+ min/max mips through glref_tex3d_level), 2: atlas as GL_R8,
 * 3: atlas as the reference's literal GL_COMPRESSED_RED.  That generic format lets the driver pick any -- or no -- compression:
 * Mesa answers with GL_COMPRESSED_RED_RGTC1 even for 3D targets (lossy: 4x4 blocks with 8 interpolated levels), a driver that
 * follows the RGTC specification (2D targets only) stores R8.  The oracle and the product implement the lossless outcome, so the
 * golden vectors are generated with kind 2; kind 3 exists to document the difference (tests/golden/make_golden_glsl.py).
 * (The reference sets the wrap modes on GL_TEXTURE_2D, i.e. not on these textures; only texelFetch is used on them.) */
int glref_tex3d(int kind, int w, int h, int d, const void* data, int max_level) {
    GLuint t; p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_3D, t);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    if (kind == 0) p_glTexImage3D(GL_TEXTURE_3D, 0, GL_RGB10_A2UI, w, h, d, 0, GL_RGBA_INTEGER, GL_UNSIGNED_INT_10_10_10_2, data);
    else if (kind == 1) p_glTexImage3D(GL_TEXTURE_3D, 0, GL_RG16F, w, h, d, 0, GL_RG, GL_HALF_FLOAT, data);
    else p_glTexImage3D(GL_TEXTURE_3D, 0, kind == 3 ? GL_COMPRESSED_RED : GL_R8, w, h, d, 0, GL_RED, GL_UNSIGNED_BYTE, data);
    p_glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_BASE_LEVEL, 0);
    p_glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MAX_LEVEL, max_level);
    if (check("glTexImage3D")) return -1;
    return (int)t;
}
int glref_tex3d_level(int tex, int level, int w, int h, int d, const void* data) {    /* range mips, renderer.cpp:186-197 */
    p_glBindTexture(GL_TEXTURE_3D, (GLuint)tex);
    p_glTexImage3D(GL_TEXTURE_3D, level, GL_RG16F, w, h, d, 0, GL_RG, GL_HALF_FLOAT, data);
    return check("glTexImage3D level");
}
int glref_tex_internal_format(int tex, int is3d) {
    GLint f = 0;
    p_glBindTexture(is3d ? GL_TEXTURE_3D : GL_TEXTURE_2D, (GLuint)tex);
    p_glGetTexLevelParameteriv(is3d ? GL_TEXTURE_3D : GL_TEXTURE_2D, 0, GL_TEXTURE_INTERNAL_FORMAT, &f);
    return (int)f;
}
/* environment map: RGB32F, GL_LINEAR, GL_REPEAT in s, GL_CLAMP_TO_EDGE in t (cppgl's Texture2D defaults are not vendored: the same
 * assumption as oracle/volren_oracle.c orc_env_texture) */
int glref_tex2d_rgb32f(int w, int h, const float* rgb) {
    GLuint t; p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB32F, w, h, 0, GL_RGB, GL_FLOAT, rgb);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    if (check("glTexImage2D rgb32f")) return -1;
    return (int)t;
}
/* empty single-channel / four-channel float textures (importance map: environment.cpp:15; colour image: renderer.cpp:12-14) */
int glref_tex2d_empty(int w, int h, int channels) {
    GLuint t; p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_2D, t);
    float* zero = (float*)calloc((size_t)w * h * channels, sizeof(float));
    p_glTexImage2D(GL_TEXTURE_2D, 0, channels == 1 ? GL_R32F : GL_RGBA32F, w, h, 0, channels == 1 ? GL_RED : GL_RGBA, GL_FLOAT, zero);
    free(zero);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, channels == 1 ? GL_NEAREST_MIPMAP_NEAREST : GL_NEAREST);
    if (check("glTexImage2D empty")) return -1;
    return (int)t;
}
int glref_upload_tex2d(int tex, int w, int h, const float* rgba) {
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, rgba);
    return check("glTexImage2D upload");
}
int glref_bind_image(int unit, int tex, int channels, int access) {        /* access 0 read-write, 1 write-only */
    p_glBindImageTexture((GLuint)unit, (GLuint)tex, 0, GL_FALSE, 0, access ? GL_WRITE_ONLY : GL_READ_WRITE, channels == 1 ? GL_R32F : GL_RGBA32F);
    return check("glBindImageTexture");
}
int glref_generate_mipmap(int tex) {
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glGenerateMipmap(GL_TEXTURE_2D);
    return check("glGenerateMipmap");
}
int glref_read_tex2d(int tex, int level, int channels, float* out) {
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glGetTexImage(GL_TEXTURE_2D, level, channels == 1 ? GL_RED : GL_RGBA, GL_FLOAT, out);
    return check("glGetTexImage");
}

/* ---------------------------------------------------------------- buffers */
int glref_ssbo(int binding, const void* data, long bytes) {               /* std430 buffer at `binding` (LUT: binding 4) */
    GLuint b; p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    p_glBufferData(GL_SHADER_STORAGE_BUFFER, bytes, data, GL_DYNAMIC_COPY);
    p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)binding, b);
    if (check("ssbo")) return -1;
    return (int)b;
}
int glref_read_ssbo(int buf, void* out, long bytes) {
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, (GLuint)buf);
    p_glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, bytes, out);
    return check("glGetBufferSubData");
}

/* ---------------------------------------------------------------- dispatch */
int glref_dispatch(int gx, int gy, int gz) {
    p_glDispatchCompute((GLuint)gx, (GLuint)gy, (GLuint)gz);
    p_glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    return check("glDispatchCompute");
}

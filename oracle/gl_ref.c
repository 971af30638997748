// TEST INFRASTRUCTURE (oracle side). Not part of the product path; runs only in the build container.
//
// gl_ref -- executes the reference's OWN GLSL (vertex + fragment shader of index.js:77-175, handed over as text files that
// oracle/gen_golden_gl.js extracts from /root/reference/index.js at run time -- never stored in this repository) on Mesa's
// software rasteriser (llvmpipe), with the draw state the reference's three.js material asks for (index.js:176-181):
// instanced non-indexed quads, two data textures, CustomBlending with blendSrcAlpha = One, depthTest LEQUAL, depthWrite off.
// Its images are the pin for the GPU half of the path ("GPU half: no" -- there is no X server, EGL or OSMesa
// in the image, but swrast_dri.so can be driven directly through the DRI swrast loader interface of
// <GL/internal/dri_interface.h>, which is what libGL and the X server do).
//
// What is build-authored here (the reference gets it from three.js r147, which is not in /root/reference): the program
// prefix three.js puts in front of a ShaderMaterial's text on WebGL2 (#version, attribute/varying/gl_FragColor defines,
// default precisions, the built-in `position` attribute) [3p-memory], DataTexture's default NEAREST filters, and the
// CustomBlending defaults (src SRC_ALPHA, dst ONE_MINUS_SRC_ALPHA, equation ADD; dstAlpha follows dst).
//
//   gl_ref <job dir> : reads job.txt (key value lines) + vs.glsl fs.glsl positions.bin index.bin cs.bin cc.bin,
//                      writes out_rgba8.bin (RGBA8 framebuffer: per-fragment unorm8 rounding like WebGL's default
//                      framebuffer), out_float.bin (RGBA32F framebuffer, 4 floats per pixel: the same shading, raster and
//                      blend without the intermediate rounding), both with row 0 = top, and out.txt (fragments that passed).
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static int g_w = 64, g_h = 64;
static void cb_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void)d; (void)p; *x = 0; *y = 0; *w = g_w; *h = g_h; }
static void cb_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void cb_put2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void cb_get2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }
static const __DRIswrastLoaderExtension g_loader = { .base = { __DRI_SWRAST_LOADER, 3 }, .getDrawableInfo = cb_info, .putImage = cb_put,
                                                     .getImage = cb_get, .putImage2 = cb_put2, .getImage2 = cb_get2 };
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

static void *(*gpa)(const char *);
#define GLF(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
GLF(const GLubyte *, glGetString, GLenum); GLF(GLenum, glGetError, void); GLF(void, glGetIntegerv, GLenum, GLint *);
GLF(GLuint, glCreateShader, GLenum); GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLF(void, glCompileShader, GLuint); GLF(void, glGetShaderiv, GLuint, GLenum, GLint *); GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(GLuint, glCreateProgram, void); GLF(void, glAttachShader, GLuint, GLuint); GLF(void, glLinkProgram, GLuint);
GLF(void, glGetProgramiv, GLuint, GLenum, GLint *); GLF(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *); GLF(void, glUseProgram, GLuint);
GLF(GLint, glGetAttribLocation, GLuint, const GLchar *); GLF(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLF(void, glUniform1i, GLint, GLint); GLF(void, glUniform1f, GLint, GLfloat); GLF(void, glUniform2f, GLint, GLfloat, GLfloat);
GLF(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *);
GLF(void, glGenVertexArrays, GLsizei, GLuint *); GLF(void, glBindVertexArray, GLuint); GLF(void, glGenBuffers, GLsizei, GLuint *);
GLF(void, glBindBuffer, GLenum, GLuint); GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLF(void, glEnableVertexAttribArray, GLuint); GLF(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
GLF(void, glVertexAttribIPointer, GLuint, GLint, GLenum, GLsizei, const void *); GLF(void, glVertexAttribDivisor, GLuint, GLuint);
GLF(void, glGenTextures, GLsizei, GLuint *); GLF(void, glBindTexture, GLenum, GLuint); GLF(void, glActiveTexture, GLenum);
GLF(void, glTexParameteri, GLenum, GLenum, GLint); GLF(void, glPixelStorei, GLenum, GLint);
GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
GLF(void, glGenFramebuffers, GLsizei, GLuint *); GLF(void, glBindFramebuffer, GLenum, GLuint);
GLF(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint); GLF(GLenum, glCheckFramebufferStatus, GLenum);
GLF(void, glViewport, GLint, GLint, GLsizei, GLsizei); GLF(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat); GLF(void, glClearDepth, GLdouble);
GLF(void, glClear, GLbitfield); GLF(void, glEnable, GLenum); GLF(void, glDisable, GLenum); GLF(void, glDepthFunc, GLenum); GLF(void, glDepthMask, GLboolean);
GLF(void, glBlendEquation, GLenum); GLF(void, glBlendFuncSeparate, GLenum, GLenum, GLenum, GLenum);
GLF(void, glDrawArraysInstanced, GLenum, GLint, GLsizei, GLsizei); GLF(void, glFinish, void);
GLF(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
GLF(void, glGenQueries, GLsizei, GLuint *); GLF(void, glBeginQuery, GLenum, GLuint); GLF(void, glEndQuery, GLenum);
GLF(void, glGetQueryObjectui64v, GLuint, GLenum, GLuint64 *);
GLF(void, glCullFace, GLenum); GLF(void, glFrontFace, GLenum); GLF(void, glScissor, GLint, GLint, GLsizei, GLsizei);
#define LOAD(name) do { *(void **)(&p_##name) = gpa(#name); if (!p_##name) { fprintf(stderr, "gl_ref: no %s\n", #name); return 2; } } while (0)

static char *slurp(const char *dir, const char *name, size_t *len)
{
    char path[4096]; snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb"); if (!f) { fprintf(stderr, "gl_ref: cannot read %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = malloc((size_t)n + 1); if (fread(b, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "gl_ref: short read %s\n", path); exit(2); }
    b[n] = 0; fclose(f); if (len) *len = (size_t)n; return b;
}
static void spill(const char *dir, const char *name, const void *p, size_t n)
{
    char path[4096]; snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "wb"); if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "gl_ref: cannot write %s\n", path); exit(2); } fclose(f);
}

// what three.js' WebGLProgram prepends to a ShaderMaterial on a WebGL2 context (the parts a shader can observe) [3p-memory]
static const char *VS_PREFIX =
    "#version 300 es\n#define attribute in\n#define varying out\n#define texture2D texture\nprecision highp float;\nprecision highp int;\n"
    "uniform mat4 modelMatrix;\nuniform mat4 modelViewMatrix;\nuniform mat4 projectionMatrix;\nuniform mat4 viewMatrix;\nuniform mat3 normalMatrix;\n"
    "uniform vec3 cameraPosition;\nuniform bool isOrthographic;\nattribute vec3 position;\nattribute vec3 normal;\nattribute vec2 uv;\n";
static const char *FS_PREFIX =
    "#version 300 es\n#define varying in\nlayout(location = 0) out highp vec4 pc_fragColor;\n#define gl_FragColor pc_fragColor\n"
    "#define gl_FragDepthEXT gl_FragDepth\n#define texture2D texture\nprecision highp float;\nprecision highp int;\n"
    "uniform mat4 viewMatrix;\nuniform vec3 cameraPosition;\nuniform bool isOrthographic;\n";

static GLuint compile(GLenum kind, const char *prefix, const char *body)
{
    GLuint s = p_glCreateShader(kind);
    const char *src[2] = { prefix, body };
    p_glShaderSource(s, 2, src, NULL); p_glCompileShader(s);
    GLint ok = 0; p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[8192]; p_glGetShaderInfoLog(s, sizeof log, NULL, log); fprintf(stderr, "gl_ref: shader compile failed:\n%s\n", log); exit(3); }
    return s;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: gl_ref <job dir>\n"); return 2; }
    const char *dir = argv[1];
    // ---- job description
    int W = 0, H = 0, texw = 0, texh = 0, count = 0, depth_test = 1, depth_write = 0, sx0 = 0, sx1 = 0;
    float viewport[2] = { 0, 0 }, focal = 0, proj[16], mv[16], clear[4] = { 0, 0, 0, 1 };
    {
        char *t = slurp(dir, "job.txt", NULL), *save = NULL;
        for (char *line = strtok_r(t, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
            char key[64]; int off = 0;
            if (sscanf(line, "%63s %n", key, &off) < 1) continue;
            const char *v = line + off;
            if (!strcmp(key, "width")) W = atoi(v); else if (!strcmp(key, "height")) H = atoi(v);
            else if (!strcmp(key, "tex_width")) texw = atoi(v); else if (!strcmp(key, "tex_height")) texh = atoi(v);
            else if (!strcmp(key, "instances")) count = atoi(v); else if (!strcmp(key, "depth_test")) depth_test = atoi(v);
            else if (!strcmp(key, "depth_write")) depth_write = atoi(v); else if (!strcmp(key, "focal")) focal = strtof(v, NULL);
            else if (!strcmp(key, "viewport")) sscanf(v, "%f %f", &viewport[0], &viewport[1]);
            else if (!strcmp(key, "strip")) sscanf(v, "%d %d", &sx0, &sx1);
            else if (!strcmp(key, "clear")) sscanf(v, "%f %f %f %f", &clear[0], &clear[1], &clear[2], &clear[3]);
            else if (!strcmp(key, "projection") || !strcmp(key, "model_view")) {
                float *m = key[0] == 'p' ? proj : mv; char *e = (char *)v;
                for (int i = 0; i < 16; i++) m[i] = strtof(e, &e);
            }
        }
        free(t);
    }
    if (W <= 0 || H <= 0 || texw <= 0 || texh <= 0 || count < 0) { fprintf(stderr, "gl_ref: bad job\n"); return 2; }
    if (sx1 <= sx0) { sx0 = 0; sx1 = W; }                           // `strip x0 x1`: only these pixel columns are drawn (scissor) and returned
    if (sx0 < 0 || sx1 > W) { fprintf(stderr, "gl_ref: bad strip\n"); return 2; }
    g_w = W; g_h = H;
    size_t n_vs, n_fs, n_pos, n_idx, n_cs, n_cc;
    char *vs = slurp(dir, "vs.glsl", &n_vs), *fs = slurp(dir, "fs.glsl", &n_fs);
    float *pos = (float *)slurp(dir, "positions.bin", &n_pos);
    uint32_t *idx = (uint32_t *)slurp(dir, "index.bin", &n_idx);
    float *cs = (float *)slurp(dir, "cs.bin", &n_cs);
    uint32_t *cc = (uint32_t *)slurp(dir, "cc.bin", &n_cc);
    // optional opaque scene under the splats: scene_depth.bin (W x H f32 window depth), scene_rgba.bin (W x H RGBA8), rows top-down
    float *scene_depth = NULL; uint8_t *scene_rgba = NULL;
    {
        char path[4096]; size_t n;
        snprintf(path, sizeof path, "%s/scene_depth.bin", dir);
        FILE *f = fopen(path, "rb"); if (f) { fclose(f); scene_depth = (float *)slurp(dir, "scene_depth.bin", &n); if (n != (size_t)W * H * 4) { fprintf(stderr, "gl_ref: scene_depth size\n"); return 2; } }
        snprintf(path, sizeof path, "%s/scene_rgba.bin", dir);
        f = fopen(path, "rb"); if (f) { fclose(f); scene_rgba = (uint8_t *)slurp(dir, "scene_rgba.bin", &n); if (n != (size_t)W * H * 4) { fprintf(stderr, "gl_ref: scene_rgba size\n"); return 2; } }
    }
    if (n_cs != (size_t)texw * texh * 16 || n_cc != n_cs || n_idx < (size_t)count * 4 || n_pos % 12) { fprintf(stderr, "gl_ref: array sizes do not match the job\n"); return 2; }

    // ---- a GL context on Mesa's software rasteriser, straight through the DRI swrast interface
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "gl_ref: %s\n", dlerror()); return 4; }
    const __DRIextension **(*get)(void) = dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get) { fprintf(stderr, "gl_ref: no swrast driver entry\n"); return 4; }
    const __DRIextension **exts = get();
    const __DRIcoreExtension *core = NULL; const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) { fprintf(stderr, "gl_ref: driver lacks DRI_Core / DRI_SWRast v4\n"); return 4; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!scr || !configs || !configs[0]) { fprintf(stderr, "gl_ref: createNewScreen2 failed\n"); return 4; }
    unsigned err = 0;
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    __DRIdrawable *dr = ctx ? sw->createNewDrawable(scr, configs[0], NULL) : NULL;
    if (!ctx || !dr || !core->bindContext(ctx, dr, dr)) { fprintf(stderr, "gl_ref: no context (error %u)\n", err); return 4; }
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    gpa = glapi ? (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address") : NULL;
    if (!gpa) { fprintf(stderr, "gl_ref: no _glapi_get_proc_address\n"); return 4; }
    LOAD(glGetString); LOAD(glGetError); LOAD(glGetIntegerv); LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader); LOAD(glGetShaderiv);
    LOAD(glGetShaderInfoLog); LOAD(glCreateProgram); LOAD(glAttachShader); LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog);
    LOAD(glUseProgram); LOAD(glGetAttribLocation); LOAD(glGetUniformLocation); LOAD(glUniform1i); LOAD(glUniform1f); LOAD(glUniform2f);
    LOAD(glUniformMatrix4fv); LOAD(glGenVertexArrays); LOAD(glBindVertexArray); LOAD(glGenBuffers); LOAD(glBindBuffer); LOAD(glBufferData);
    LOAD(glEnableVertexAttribArray); LOAD(glVertexAttribPointer); LOAD(glVertexAttribIPointer); LOAD(glVertexAttribDivisor); LOAD(glGenTextures);
    LOAD(glBindTexture); LOAD(glActiveTexture); LOAD(glTexParameteri); LOAD(glPixelStorei); LOAD(glTexImage2D); LOAD(glGenFramebuffers);
    LOAD(glBindFramebuffer); LOAD(glFramebufferTexture2D); LOAD(glCheckFramebufferStatus); LOAD(glViewport); LOAD(glClearColor); LOAD(glClearDepth);
    LOAD(glClear); LOAD(glEnable); LOAD(glDisable); LOAD(glDepthFunc); LOAD(glDepthMask); LOAD(glBlendEquation); LOAD(glBlendFuncSeparate);
    LOAD(glDrawArraysInstanced); LOAD(glFinish); LOAD(glReadPixels); LOAD(glGenQueries); LOAD(glBeginQuery); LOAD(glEndQuery); LOAD(glGetQueryObjectui64v); LOAD(glCullFace); LOAD(glFrontFace); LOAD(glScissor);
    GLint maxtex = 0; p_glGetIntegerv(GL_MAX_TEXTURE_SIZE, &maxtex);
    if (texw > maxtex || texh > maxtex) { fprintf(stderr, "gl_ref: texture %dx%d exceeds %d\n", texw, texh, maxtex); return 2; }

    // ---- program = three.js prefix + the reference's shader text
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, compile(GL_VERTEX_SHADER, VS_PREFIX, vs)); p_glAttachShader(prog, compile(GL_FRAGMENT_SHADER, FS_PREFIX, fs));
    p_glLinkProgram(prog);
    GLint ok = 0; p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { char log[8192]; p_glGetProgramInfoLog(prog, sizeof log, NULL, log); fprintf(stderr, "gl_ref: link failed:\n%s\n", log); return 3; }
    p_glUseProgram(prog);

    // ---- geometry: the quad's vertices as the reference filled them, the sorted index list as the instanced attribute
    GLuint vao, vbo[2];
    p_glGenVertexArrays(1, &vao); p_glBindVertexArray(vao); p_glGenBuffers(2, vbo);
    const GLint a_pos = p_glGetAttribLocation(prog, "position"), a_idx = p_glGetAttribLocation(prog, "splatIndex");
    if (a_pos < 0 || a_idx < 0) { fprintf(stderr, "gl_ref: attributes position / splatIndex not active (%d, %d)\n", a_pos, a_idx); return 3; }
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo[0]); p_glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)n_pos, pos, GL_STATIC_DRAW);
    p_glEnableVertexAttribArray((GLuint)a_pos); p_glVertexAttribPointer((GLuint)a_pos, 3, GL_FLOAT, GL_FALSE, 0, NULL);
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo[1]); p_glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)((size_t)count * 4 + 4), idx, GL_DYNAMIC_DRAW);
    p_glEnableVertexAttribArray((GLuint)a_idx); p_glVertexAttribIPointer((GLuint)a_idx, 1, GL_UNSIGNED_INT, 0, NULL); p_glVertexAttribDivisor((GLuint)a_idx, 1);

    // ---- the two data textures (DataTexture: NEAREST, no mipmaps, no flip)
    GLuint tex[2]; p_glGenTextures(2, tex); p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    for (int t = 0; t < 2; t++) {
        p_glActiveTexture(GL_TEXTURE0 + (GLenum)t); p_glBindTexture(GL_TEXTURE_2D, tex[t]);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST); p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE); p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        if (t == 0) p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, texw, texh, 0, GL_RGBA, GL_FLOAT, cs);
        else p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32UI, texw, texh, 0, GL_RGBA_INTEGER, GL_UNSIGNED_INT, cc);
    }
    p_glUniform1i(p_glGetUniformLocation(prog, "centerAndScaleTexture"), 0); p_glUniform1i(p_glGetUniformLocation(prog, "covAndColorTexture"), 1);
    p_glUniform2f(p_glGetUniformLocation(prog, "viewport"), viewport[0], viewport[1]); p_glUniform1f(p_glGetUniformLocation(prog, "focal"), focal);
    p_glUniformMatrix4fv(p_glGetUniformLocation(prog, "gsProjectionMatrix"), 1, GL_FALSE, proj);
    p_glUniformMatrix4fv(p_glGetUniformLocation(prog, "gsModelViewMatrix"), 1, GL_FALSE, mv);

    // ---- the material's fixed-function state
    p_glEnable(GL_BLEND); p_glBlendEquation(GL_FUNC_ADD);
    p_glBlendFuncSeparate(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA, GL_ONE, GL_ONE_MINUS_SRC_ALPHA);   // CustomBlending, blendSrcAlpha = One
    if (depth_test) { p_glEnable(GL_DEPTH_TEST); p_glDepthFunc(GL_LEQUAL); } else p_glDisable(GL_DEPTH_TEST);
    p_glDepthMask(depth_write ? GL_TRUE : GL_FALSE);
    p_glEnable(GL_CULL_FACE); p_glCullFace(GL_BACK); p_glFrontFace(GL_CCW);                          // material.side = FrontSide (three.js default)
    uint64_t passed[2] = { 0, 0 };
    double draw_s[2] = { 0, 0 };
    for (int pass = 0; pass < 2; pass++) {                        // 0: RGBA8 colour buffer, 1: RGBA32F colour buffer
        GLuint fbo, col, dep;
        p_glGenFramebuffers(1, &fbo); p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        p_glGenTextures(1, &col); p_glActiveTexture(GL_TEXTURE2); p_glBindTexture(GL_TEXTURE_2D, col);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST); p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        // colour buffer: cleared, or the opaque scene's image (what three.js drew before the transparent splat mesh)
        const size_t npx = (size_t)W * H;
        uint8_t *c8 = NULL; float *cf = NULL;
        if (scene_rgba) {                                          // job rows are top-down, GL rows bottom-up
            c8 = malloc(npx * 4); cf = malloc(npx * 16);
            for (int y = 0; y < H; y++) memcpy(c8 + (size_t)y * W * 4, scene_rgba + (size_t)(H - 1 - y) * W * 4, (size_t)W * 4);
            for (size_t i = 0; i < npx * 4; i++) cf[i] = (float)c8[i] / 255.0f;
        }
        if (pass == 0) p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, W, H, 0, GL_RGBA, GL_UNSIGNED_BYTE, c8);
        else p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, cf);
        free(c8); free(cf);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, col, 0);
        // depth buffer: cleared to the far plane, or the opaque scene's window-space depth (32-bit float attachment: the test
        // is the reference's LEQUAL without a fixed-point quantisation of its own)
        p_glGenTextures(1, &dep); p_glBindTexture(GL_TEXTURE_2D, dep);
        if (scene_depth) {
            float *df = malloc(npx * 4);
            for (int y = 0; y < H; y++) memcpy(df + (size_t)y * W, scene_depth + (size_t)(H - 1 - y) * W, (size_t)W * 4);
            p_glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT32F, W, H, 0, GL_DEPTH_COMPONENT, GL_FLOAT, df);
            free(df);
        } else p_glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT24, W, H, 0, GL_DEPTH_COMPONENT, GL_UNSIGNED_INT, NULL);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_TEXTURE_2D, dep, 0);
        if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "gl_ref: framebuffer incomplete\n"); return 3; }
        p_glViewport(0, 0, W, H);
        if (sx0 != 0 || sx1 != W) { p_glEnable(GL_SCISSOR_TEST); p_glScissor(sx0, 0, sx1 - sx0, H); }
        p_glDepthMask(GL_TRUE); p_glClearColor(clear[0], clear[1], clear[2], clear[3]); p_glClearDepth(1.0);
        p_glClear((scene_rgba ? 0 : GL_COLOR_BUFFER_BIT) | (scene_depth ? 0 : GL_DEPTH_BUFFER_BIT));
        p_glDepthMask(depth_write ? GL_TRUE : GL_FALSE);
        struct timespec t0, t1; p_glFinish(); clock_gettime(CLOCK_MONOTONIC, &t0);
        GLuint q; p_glGenQueries(1, &q); p_glBeginQuery(GL_SAMPLES_PASSED, q);
        if (count) p_glDrawArraysInstanced(GL_TRIANGLES, 0, (GLsizei)(n_pos / 12), count);
        p_glEndQuery(GL_SAMPLES_PASSED); p_glFinish();
        clock_gettime(CLOCK_MONOTONIC, &t1); draw_s[pass] = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        GLuint64 r = 0; p_glGetQueryObjectui64v(q, GL_QUERY_RESULT, &r); passed[pass] = r;
        const int SW = sx1 - sx0;
        const size_t px = (size_t)SW * H;
        if (pass == 0) {
            uint8_t *buf = malloc(px * 4), *flip = malloc(px * 4);
            p_glPixelStorei(GL_PACK_ALIGNMENT, 1); p_glReadPixels(sx0, 0, SW, H, GL_RGBA, GL_UNSIGNED_BYTE, buf);
            for (int y = 0; y < H; y++) memcpy(flip + (size_t)y * SW * 4, buf + (size_t)(H - 1 - y) * SW * 4, (size_t)SW * 4);
            spill(dir, "out_rgba8.bin", flip, px * 4); free(buf); free(flip);
        } else {
            float *buf = malloc(px * 16), *flip = malloc(px * 16);
            p_glReadPixels(sx0, 0, SW, H, GL_RGBA, GL_FLOAT, buf);
            for (int y = 0; y < H; y++) memcpy(flip + (size_t)y * SW * 4, buf + (size_t)(H - 1 - y) * SW * 4, (size_t)SW * 16);
            spill(dir, "out_float.bin", flip, px * 16); free(buf); free(flip);
        }
        const GLenum e = p_glGetError();
        if (e != GL_NO_ERROR) { fprintf(stderr, "gl_ref: GL error 0x%x\n", e); return 3; }
    }
    char out[512];
    snprintf(out, sizeof out, "fragments_rgba8 %llu\nfragments_float %llu\nrenderer %s\nversion %s\ndraw_seconds_rgba8 %.6f\ndraw_seconds_float %.6f\n",
             (unsigned long long)passed[0], (unsigned long long)passed[1], (const char *)p_glGetString(GL_RENDERER), (const char *)p_glGetString(GL_VERSION),
             draw_s[0], draw_s[1]);
    spill(dir, "out.txt", out, strlen(out));
    return 0;
}

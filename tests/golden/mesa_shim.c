/*
 * Headless desktop OpenGL for the golden-vector generators: Mesa's software rasteriser (llvmpipe, swrast_dri.so) loaded the way
 * an X server loads it in swrast mode — through the DRI_SWRast driver interface (/usr/include/GL/internal/dri_interface.h) —
 * with no X server, no EGL and no window: a 16×16 drawable whose put/get-image callbacks go nowhere; everything is rendered into
 * framebuffer objects and read back with glReadPixels.
 *
 * TEST INFRASTRUCTURE (build container only): used by tests/golden/mesa.py and the make_golden_mesa*.py scripts; nothing under
 * shaderflow_amd/, bench.py's timed region or the GPU tests loads it.
 *
 *   gcc -O2 -shared -fPIC tests/golden/mesa_shim.c -o build/mesa/libmesa_shim.so -ldl
 */
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const __DRIcoreExtension   *core;
static const __DRIswrastExtension *swrast;
static __DRIscreen   *screen;
static __DRIcontext  *context;
static __DRIdrawable *drawable;
static void *(*get_proc)(const char *);
static char failure[256];

static void drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv) { *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv) {}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv) { memset(data, 0, (size_t)w*h*4); }
static void put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *priv) {}
static void get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *priv) { memset(data, 0, (size_t)stride*h); }

static const __DRIswrastLoaderExtension loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = drawable_info, .putImage = put_image, .getImage = get_image,
    .putImage2 = put_image2, .getImage2 = get_image2,
};
static const __DRIextension *loader_extensions[] = {&loader.base, NULL};

const char *mesa_error(void) { return failure; }

/* creates the screen, an OpenGL `major.minor` core-profile context and makes it current; 0 on success */
int mesa_init(const char *driver_path, const char *glapi_path, int major, int minor) {
    if (context) return 0;
    void *glapi = dlopen(glapi_path, RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) { snprintf(failure, sizeof failure, "dlopen glapi: %s", dlerror()); return 1; }
    void *driver = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
    if (!driver) { snprintf(failure, sizeof failure, "dlopen driver: %s", dlerror()); return 2; }
    get_proc = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    const __DRIextension **(*get_extensions)(void) = (const __DRIextension **(*)(void))dlsym(driver, "__driDriverGetExtensions_swrast");
    if (!get_proc || !get_extensions) { snprintf(failure, sizeof failure, "missing entry points"); return 3; }
    const __DRIextension **extensions = get_extensions();
    for (int i = 0; extensions[i]; i++) {
        if (!strcmp(extensions[i]->name, __DRI_CORE))   core   = (const __DRIcoreExtension *)extensions[i];
        if (!strcmp(extensions[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)extensions[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { snprintf(failure, sizeof failure, "driver lacks DRI_Core / DRI_SWRast v4"); return 4; }
    const __DRIconfig **configs = NULL;
    screen = swrast->createNewScreen2(0, loader_extensions, extensions, &configs, NULL);
    if (!screen || !configs || !configs[0]) { snprintf(failure, sizeof failure, "createNewScreen2 failed"); return 5; }
    /* an RGBA8888 config without depth is all the default drawable needs: every render goes to an FBO */
    const __DRIconfig *config = configs[0];
    for (int i = 0; configs[i]; i++) {
        unsigned red = 0, alpha = 0, dbl = 1;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &red);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &alpha);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &dbl);
        if (red == 8 && alpha == 8 && !dbl) { config = configs[i]; break; }
    }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)major, __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t)minor};
    unsigned error = 0;
    context = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, config, NULL, 2, attribs, &error, NULL);
    if (!context) { snprintf(failure, sizeof failure, "createContextAttribs(core %d.%d) failed: %u", major, minor, error); return 6; }
    drawable = swrast->createNewDrawable(screen, config, NULL);
    if (!drawable) { snprintf(failure, sizeof failure, "createNewDrawable failed"); return 7; }
    if (!core->bindContext(context, drawable, drawable)) { snprintf(failure, sizeof failure, "bindContext failed"); return 8; }
    return 0;
}

void *mesa_proc(const char *name) { return get_proc ? get_proc(name) : NULL; }

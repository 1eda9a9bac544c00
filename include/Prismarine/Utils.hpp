#pragma once
// Prismarine/Utils.hpp -- GL-free replacement of the reference's GL utility shim
// (reference Include/Prismarine/Utils.hpp: loadShader, allocateBuffer<T>, dispatch, :28-178).
// `GLuint` stays an opaque 32-bit buffer name so signatures such as RadixSort::sort(GLuint&, GLuint&, ...)
// and TriangleArrayInstance::setVertices(const GLuint&) keep compiling; a name is a psm_buf handle of
// the process-wide context (the reference's single GL context, Viewer.cpp:371).

#define RAY_TRACING_ENGINE

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "../psm_hip.h"
#include "psm_glm.hpp"

#define NSM psm

typedef uint32_t GLuint;
typedef int32_t GLint;
typedef float GLfloat;

namespace NSM {

    class BaseClass {};
    class Pipeline;
    class TriangleHierarchy;

    static inline int32_t tiled(int32_t sz, int32_t gmaxtile) {  // Utils.hpp:37-39
        return (int32_t)std::ceil((double)sz / (double)gmaxtile);
    }

    template<class T>
    size_t strided(size_t sizeo) { return sizeof(T) * sizeo; }

    // Like a GL context, one psm context is "current": objects are created in it. ContextScope makes another
    // one current for a while (FrameBatch creates each lane's objects in the lane's own context).
    inline psm_ctx *& currentContextOverride() { static thread_local psm_ctx * c = nullptr; return c; }
    struct ContextScope {
        psm_ctx * prev;
        explicit ContextScope(psm_ctx * c) : prev(currentContextOverride()) { currentContextOverride() = c; }
        ~ContextScope() { currentContextOverride() = prev; }
    };

    // the process-wide context: device from PSM_DEVICE (default 0); created on first use
    inline psm_ctx * context() {
        if (currentContextOverride()) return currentContextOverride();
        static psm_ctx * ctx = nullptr;
        if (!ctx) {
            const char * e = std::getenv("PSM_DEVICE");
            int rc = psm_ctx_create(e ? std::atoi(e) : 0, &ctx);
            if (rc != PSM_OK) {
                std::cerr << "psm: no gfx950 device (psm_ctx_create -> " << rc << "); there is no CPU fallback" << std::endl;
                std::abort();
            }
        }
        return ctx;
    }

    // errors go to stderr and execution continues, like the reference (Utils.hpp:108-120)
    inline void check(int rc, const char * what) {
        if (rc != PSM_OK) std::cerr << "psm: " << what << " failed (" << rc << "): " << psm_last_error(context()) << std::endl;
    }

    // allocateBuffer<T>(count), Utils.hpp:140-150
    template<class T>
    inline GLuint allocateBuffer(size_t count) {
        GLuint h = 0;
        check(psm_buf_alloc(context(), strided<T>(count), &h), "allocateBuffer");
        return h;
    }
    inline void deleteBuffer(GLuint & h) { if (h != 0 && h != GLuint(-1)) psm_buf_free(context(), h); h = GLuint(-1); }
    inline void bufferSubData(GLuint h, size_t offset, size_t bytes, const void * src) { check(psm_buf_upload(context(), h, offset, src, bytes), "bufferSubData"); }
    inline void getBufferSubData(GLuint h, size_t offset, size_t bytes, void * dst) { check(psm_buf_download(context(), h, offset, dst, bytes), "getBufferSubData"); }
}

// Prints the camera matrices the GL-free host layer computes with its OWN glm stand-in (psm_glm.hpp built with
// -DPSM_NO_SYSTEM_GLM) for the cases given on stdin; tests/test_oracle_cpu.py compares them bit for bit with
// tests/golden/glm_host_formulas.npz (produced by the reference's vendored glm).
// stdin: n, then n lines "ex ey ez vx vy vz width height" (floats as hex bit patterns); stdout: 32 hex words per case.
// With the argument `gltf`: the viewer's node transforms (Source/Examples/Viewer.cpp:246-258) for the cases of
// tests/golden/glm_gltf_transforms.npz. stdin: n, then per case "has_matrix has_t has_s has_r" and 16 + 16 + 3 + 3 + 4 doubles
// (parent, matrix, translation, scale, rotation; 64-bit hex patterns); stdout: 16 doubles (transform) + 32 floats (what
// setTransform uploads: transpose(t), inverse(t)) as hex words.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "Prismarine/psm_glm.hpp"

static float f(uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; }
static uint32_t u(float v) { uint32_t x; std::memcpy(&x, &v, 4); return x; }

static double d(unsigned long long v) { double x; std::memcpy(&x, &v, 8); return x; }
static unsigned long long ud(double v) { unsigned long long x; std::memcpy(&x, &v, 8); return x; }

static int gltf_cases() {
    int n = 0;
    if (std::scanf("%d", &n) != 1) return 1;
    for (int i = 0; i < n; i++) {
        int has[4];
        double v[42];
        if (std::scanf("%d %d %d %d", &has[0], &has[1], &has[2], &has[3]) != 4) return 1;
        for (int k = 0; k < 42; k++) { unsigned long long w; if (std::scanf("%llx", &w) != 1) return 1; v[k] = d(w); }
        const double *parent = v, *matrix = v + 16, *tr = v + 32, *sc = v + 35, *ro = v + 38;
        // Viewer.cpp:246-253 (tinygltf reads T / R / S only where the node has no matrix)
        glm::dmat4 inTransform = glm::make_mat4(parent);
        glm::dmat4 localTransform(1.0);
        localTransform *= (has[0] ? glm::make_mat4(matrix) : glm::dmat4(1.0));
        localTransform *= (has[1] && !has[0] ? glm::translate(glm::make_vec3(tr)) : glm::dmat4(1.0));
        localTransform *= (has[2] && !has[0] ? glm::scale(glm::make_vec3(sc)) : glm::dmat4(1.0));
        localTransform *= (has[3] && !has[0] ? glm::mat4_cast(glm::make_quat(ro)) : glm::dmat4(1.0));
        glm::dmat4 transform = inTransform * localTransform;
        // VertexInstance.inl:54-58 behind setTransform(mat4(transform))
        glm::mat4 t = glm::mat4(transform);
        glm::mat4 a = glm::transpose(t), b = glm::inverse(t);
        const double* o = glm::value_ptr(transform);
        for (int k = 0; k < 16; k++) std::printf("%016llx ", ud(o[k]));
        for (int k = 0; k < 16; k++) std::printf("%08x ", u(glm::value_ptr(a)[k]));
        for (int k = 0; k < 16; k++) std::printf("%08x ", u(glm::value_ptr(b)[k]));
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "gltf") == 0) return gltf_cases();
    int n = 0;
    if (std::scanf("%d", &n) != 1) return 1;
    for (int i = 0; i < n; i++) {
        uint32_t e[6]; int w, h;
        if (std::scanf("%x %x %x %x %x %x %d %d", &e[0], &e[1], &e[2], &e[3], &e[4], &e[5], &w, &h) != 8) return 1;
        glm::vec3 eye(f(e[0]), f(e[1]), f(e[2])), view(f(e[3]), f(e[4]), f(e[5]));
        // Pipeline.inl:279-312 as include/Prismarine/Pipeline.inl evaluates it
        glm::mat4 persp = glm::perspective(glm::pi<float>() / 3.0f, float(w) / float(h), 0.001f, 1000.0f);
        glm::mat4 side = glm::lookAt(eye, view, glm::vec3(0.0f, 1.0f, 0.0f));
        glm::mat4 ci = glm::transpose(glm::inverse(side)), pi = glm::transpose(glm::inverse(persp));
        const float* a = glm::value_ptr(ci);
        const float* b = glm::value_ptr(pi);
        for (int k = 0; k < 16; k++) std::printf("%08x ", u(a[k]));
        for (int k = 0; k < 16; k++) std::printf("%08x ", u(b[k]));
        std::printf("\n");
    }
    return 0;
}

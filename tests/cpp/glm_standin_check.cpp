// Prints the camera matrices the GL-free host layer computes with its OWN glm stand-in (psm_glm.hpp built with
// -DPSM_NO_SYSTEM_GLM) for the cases given on stdin; tests/test_oracle_cpu.py compares them bit for bit with
// tests/golden/glm_host_formulas.npz (produced by the reference's vendored glm).
// stdin: n, then n lines "ex ey ez vx vy vz width height" (floats as hex bit patterns); stdout: 32 hex words per case.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "Prismarine/psm_glm.hpp"

static float f(uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; }
static uint32_t u(float v) { uint32_t x; std::memcpy(&x, &v, 4); return x; }

int main() {
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

#include "Pipeline.hpp"

// Implementation of psm::Pipeline over the C ABI. Replaces the GL orchestration of the reference's
// Include/Prismarine/Pipeline.inl (init :60-136, resizeBuffers :174-214, camera :279-312,
// reloadQueuedRays :325-359, intersection :385-405, applyMaterials :407-421, shade :423-436,
// sample :251-277, snapHdr :439-456).

namespace NSM {

    inline Pipeline::~Pipeline() { if (rt) psm_rt_destroy(rt); }

    inline void Pipeline::init() {
        check(psm_rt_create(context(), &rt), "Pipeline::init");
        for (int i = 0; i < 6; i++) {   // Pipeline.inl:93-98
            lightColor[i] = glm::vec4((255.f / 255.f) * 150.f, (250.f / 255.f) * 150.f, (244.f / 255.f) * 150.f, 40.0f);
            lightAmbient[i] = glm::vec4(0.0f);
            lightVector[i] = glm::vec4(0.3f, 1.0f, 0.1f, 400.0f);
            lightOffset[i] = glm::vec4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        lightcount = 1;
        resizeBuffers(width, height);
        resize(displayWidth, displayHeight);
        syncUniforms();
    }

    inline void Pipeline::setLightCount(size_t lc) { lightcount = (uint32_t)(lc < 1 ? 1 : (lc > 6 ? 6 : lc)); }

    inline void Pipeline::switchMode() {   // :128-132
        clearRays(); clearSampler();
        enable360 = enable360 == 1 ? 0 : 1;
        check(psm_rt_set_camera_mode(rt, enable360), "Pipeline::switchMode");
    }

    inline void Pipeline::resize(const uint32_t & w, const uint32_t & h) {
        displayWidth = w; displayHeight = h;
        check(psm_rt_resize(rt, w, h), "Pipeline::resize");
    }

    inline void Pipeline::resizeBuffers(const uint32_t & w, const uint32_t & h) {
        width = w; height = h;
        check(psm_rt_resize_buffers(rt, w, h), "Pipeline::resizeBuffers");
        raycountCache = 0;
    }

    inline void Pipeline::syncUniforms() {   // Pipeline.inl:216-230
        psm_light L[6];
        for (uint32_t i = 0; i < lightcount; i++) {
            std::memcpy(L[i].lightColor, glm::value_ptr(lightColor[i]), 16);
            std::memcpy(L[i].lightVector, glm::value_ptr(lightVector[i]), 16);
            std::memcpy(L[i].lightOffset, glm::value_ptr(lightOffset[i]), 16);
            std::memcpy(L[i].lightAmbient, glm::value_ptr(lightAmbient[i]), 16);
        }
        check(psm_rt_set_lights(rt, L, lightcount), "Pipeline::syncUniforms");
    }

    inline void Pipeline::clearRays() { raycountCache = 0; }

    inline void Pipeline::reloadQueuedRays(bool, bool) {   // Pipeline.inl:325-359: raycountCache <- At
        int32_t n = 0;
        check(psm_rt_ray_count(rt, &n), "Pipeline::reloadQueuedRays");
        raycountCache = n;
    }

    inline void Pipeline::sample() { check(psm_rt_sample(rt), "Pipeline::sample"); }

    inline void Pipeline::camera(const glm::mat4 &persp, const glm::mat4 &frontSide) {   // :279-296
        glm::mat4 ci = glm::inverse(frontSide), pi = glm::inverse(persp);
        float camInv[16], projInv[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { camInv[4 * r + c] = ci[c][r]; projInv[4 * r + c] = pi[c][r]; }
        this->syncUniforms();
        check(psm_rt_camera(rt, camInv, projInv, nextRand()), "Pipeline::camera");
        reloadQueuedRays(true);
    }

    inline void Pipeline::camera(const glm::vec3 &eye, const glm::vec3 &view, const glm::mat4 &persp) {
        glm::mat4 sidemat = glm::lookAt(eye, view, glm::vec3(0.0f, 1.0f, 0.0f));
        this->camera(persp, sidemat);
    }

    inline void Pipeline::camera(const glm::vec3 &eye, const glm::vec3 &view) {   // :310-312
        this->camera(eye, view, glm::perspective(glm::pi<float>() / 3.0f, float(displayWidth) / float(displayHeight), 0.001f, 1000.0f));
    }

    inline void Pipeline::clearSampler() { check(psm_rt_clear_sampler(rt), "Pipeline::clearSampler"); }

    inline void Pipeline::reclaim() {}   // a no-op in the reference as well (:361-369)

    inline void Pipeline::render() {}    // display quad (render.vert/frag): out of scope, see snapHdr()

    inline int Pipeline::intersection(TriangleHierarchy * obj, const int clearDepth) {   // :385-405
        (void)clearDepth;
        if (!obj || obj->triangleCount <= 0) return 0;
        int32_t rsize = getRayCount();
        if (rsize <= 0) return 0;
        lastObj = obj;
        int rc = psm_rt_traverse(rt, obj->handle());
        check(rc, "Pipeline::intersection");
        return rc == PSM_OK ? 1 : 0;
    }

    inline void Pipeline::applyMaterials(MaterialSet * mat) {   // :407-421 (surface.comp is fused into shade)
        if (!mat) return;
        if (mat != matOwner || mat->revision != matRevision) {
            check(psm_rt_set_materials(rt, mat->submats.data(), (uint32_t)mat->submats.size(), mat->loadOffset), "Pipeline::applyMaterials");
            matOwner = mat; matRevision = mat->revision;
        }
        TextureSet * ts = mat->texset;   // MaterialSet::bindWithContext -> TextureSet::bindWithContext, MaterialSet.inl:20-23
        if (ts && (ts != texOwner || ts->revision != texRevision)) {
            for (uint32_t i = 1; i < 32; i++) {
                GLuint name = i < ts->textures.size() ? ts->textures[i] : GLuint(-1);
                if (name != GLuint(-1) && name < hostTextures().size()) {
                    const HostTexture & h = hostTextures()[name];
                    check(psm_rt_set_texture(rt, i, h.rgba8.data(), h.width, h.height), "Pipeline::applyMaterials(texture)");
                } else {
                    check(psm_rt_set_texture(rt, i, nullptr, 0, 0), "Pipeline::applyMaterials(texture)");
                }
            }
            texOwner = ts; texRevision = ts->revision;
        }
    }

    inline void Pipeline::shade() {   // :423-436
        int32_t rsize = getRayCount();
        if (rsize <= 0 || !lastObj) return;
        check(psm_rt_shade(rt, lastObj->handle(), nextRand()), "Pipeline::shade");
        reloadQueuedRays(true);
    }

    inline Pipeline::HdrImage Pipeline::snapRawHdr() {   // :439-447; the caller owns img.image
        HdrImage img;
        img.width = (int)displayWidth; img.height = (int)displayHeight;
        img.image = new GLfloat[(size_t)displayWidth * displayHeight * 4];
        check(psm_rt_snap(rt, img.image, 1), "Pipeline::snapRawHdr");
        return img;
    }

    inline Pipeline::HdrImage Pipeline::snapHdr() {   // :448-456
        HdrImage img;
        img.width = (int)displayWidth; img.height = (int)displayHeight;
        img.image = new GLfloat[(size_t)displayWidth * displayHeight * 4];
        check(psm_rt_snap(rt, img.image, 0), "Pipeline::snapHdr");
        return img;
    }

    inline int32_t Pipeline::getRayCount() {   // :459-461
        return raycountCache >= 32 ? raycountCache : 0;
    }
}

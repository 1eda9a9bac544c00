#pragma once
// Prismarine/Pipeline.hpp -- psm::Pipeline, same public surface as the reference
// (Include/Prismarine/Pipeline.hpp:84-137): the wavefront path-tracing loop.

#include "Utils.hpp"
#include "Structs.hpp"
#include "TriangleHierarchy.hpp"
#include "MaterialSet.hpp"

namespace NSM {

    class FrameBatch;

    class Pipeline : public BaseClass {
    protected:
        friend class FrameBatch;
        GLuint skybox = GLuint(-1);
        psm_rt * rt = nullptr;
        TriangleHierarchy * lastObj = nullptr;
        uint64_t matRevision = 0;
        const MaterialSet * matOwner = nullptr;
        uint64_t texRevision = 0;
        const TextureSet * texOwner = nullptr;
        uint32_t randState = 1;       // host rand() of Pipeline.inl:282,426 (MSVC CRT LCG), see setSeed
        uint32_t lightcount = 1;
        int enable360 = 0;            // cameraUniformData.enable360, Pipeline.inl:119
        void init();
        uint32_t nextRand() { randState = randState * 214013u + 2531011u; return (randState >> 16) & 0x7fffu; }

    public:
        Pipeline() { init(); }
        ~Pipeline();

        uint32_t width = 256;
        uint32_t height = 256;
        uint32_t displayWidth = 256;
        uint32_t displayHeight = 256;

        void setSkybox(GLuint skb) { skybox = skb; }   // kept for source compatibility; the image comes through setSkyboxImage
        // equirect RGBA8 image (public/environment.glsl:23-26; Application.hpp:46-54 upload)
        void setSkyboxImage(const uint8_t * rgba8, uint32_t w, uint32_t h) { check(psm_rt_set_skybox(rt, rgba8, w, h), "Pipeline::setSkyboxImage"); }

        void switchMode();
        void resize(const uint32_t & w, const uint32_t & h);
        void resizeBuffers(const uint32_t & w, const uint32_t & h);
        void syncUniforms();
        void reloadQueuedRays(bool doSort = false, bool sortMortons = false);

        int32_t raycountCache = 0;
        int32_t qraycountCache = 0;

        glm::vec4 lightColor[6];
        glm::vec4 lightAmbient[6];
        glm::vec4 lightVector[6];
        glm::vec4 lightOffset[6];

        struct HdrImage {
            GLfloat * image = nullptr;
            int width = 1;
            int height = 1;
        };

        HdrImage snapHdr();
        HdrImage snapRawHdr();

        void setLightCount(size_t lightcount);
        void bindUniforms() {}
        void bind() {}
        void clearRays();
        void sample();
        void camera(const glm::mat4 &persp, const glm::mat4 &frontSide);
        void camera(const glm::vec3 &eye, const glm::vec3 &view, const glm::mat4 &persp);
        void camera(const glm::vec3 &eye, const glm::vec3 &view);
        void clearSampler();
        void reclaim();
        void render();
        int intersection(TriangleHierarchy * obj, const int clearDepth = 0);
        void shade();
        void applyMaterials(MaterialSet * mat);
        int32_t getRayCount();

        // additions (no reference counterpart): explicit RNG seed, tile sharding, constant sky
        void setSeed(uint32_t seed) { randState = seed; }
        void setTile(uint32_t y0, uint32_t y1) { check(psm_rt_set_tile(rt, y0, y1), "Pipeline::setTile"); }
        void setSky(const glm::vec4 & rgba) { check(psm_rt_set_sky(rt, glm::value_ptr(rgba)), "Pipeline::setSky"); }
        psm_rt * handle() { return rt; }
        void noteTraced(TriangleHierarchy * obj) { lastObj = obj; raycountCache = 0; }   // after psm_lanes_render: the frame is over
    };
}

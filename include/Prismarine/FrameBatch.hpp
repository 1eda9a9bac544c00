#pragma once
// Prismarine/FrameBatch.hpp -- several frames in flight (new; no counterpart in the reference).
//
// GltfViewer::process() (Source/Examples/Viewer.cpp:296-312) renders one sample per pixel per call and the
// sampler accumulates the calls. A bounce round ends with its slowest ray, so one frame at a time leaves most
// of an MI355X idle for a good part of every round; FrameBatch runs `lanes` process() calls at once, each in
// its own context (HIP stream) with its own TriangleHierarchy and Pipeline, through psm_lanes_render, and
// issues their sample() in frame order on the accumulating Pipeline: the image equals the frames rendered one
// after another. rand(): the accumulating Pipeline's stream hands every frame one draw, which seeds the
// frame's own stream (camera + one draw per shade, Pipeline.inl:282,426).
//
//     psm::FrameBatch batch(4, 1920, 1080);
//     batch.allocate(1024 * 1024);
//     batch.loadMesh(mesh);                        // every lane loads the scene
//     batch.applyMaterials(materialManager);
//     batch.render(4, cam->eye, cam->view, 16);    // = 4 x process(): one 4-spp frame
//     auto img = batch.accumulator()->snapHdr();

#include "TriangleHierarchy.hpp"
#include "Pipeline.hpp"

namespace NSM {
    class FrameBatch : public BaseClass {
    protected:
        std::vector<psm_ctx *> ctxs;
        std::vector<TriangleHierarchy *> objs;
        std::vector<Pipeline *> rays;
        Pipeline * master = nullptr;
        uint32_t width = 0, height = 0;

    public:
        std::vector<psm_lane_result> lastResults;   // per frame of the last render(): rounds, rays traced

        FrameBatch(uint32_t lanes, uint32_t w, uint32_t h) : width(w), height(h) {
            master = new Pipeline();                 // in the current context: it only samples
            master->resizeBuffers(w, h);
            master->resize(w, h);
            int device = 0;
            if (const char * e = std::getenv("PSM_DEVICE")) device = std::atoi(e);
            for (uint32_t s = 0; s < lanes; s++) {
                psm_ctx * c = nullptr;
                check(psm_ctx_create(device, &c), "FrameBatch: psm_ctx_create");
                ctxs.push_back(c);
                ContextScope scope(c);
                objs.push_back(new TriangleHierarchy());
                Pipeline * p = new Pipeline();
                p->resizeBuffers(w, h);
                p->resize(w, h);
                rays.push_back(p);
            }
        }
        ~FrameBatch() {
            for (size_t s = 0; s < ctxs.size(); s++) { delete rays[s]; delete objs[s]; psm_ctx_destroy(ctxs[s]); }
            delete master;
        }
        FrameBatch(const FrameBatch &) = delete;
        FrameBatch & operator=(const FrameBatch &) = delete;

        size_t lanes() const { return ctxs.size(); }
        Pipeline * accumulator() { return master; }
        Pipeline * lane(size_t s) { return rays[s]; }
        TriangleHierarchy * hierarchy(size_t s) { return objs[s]; }

        // the scene, once per lane (vertex buffers stay in the application's context; lanes only read them)
        void allocate(const size_t & count) { for (size_t s = 0; s < objs.size(); s++) { ContextScope scope(ctxs[s]); objs[s]->allocate(count); } }
        void clearTribuffer() { for (auto o : objs) o->clearTribuffer(); }
        void loadMesh(TriangleArrayInstance * gobject) { for (auto o : objs) o->loadMesh(gobject); }
        void loadTriangles(const float * positions, const float * normals, const int32_t * materials, size_t count, const float * texcoords = nullptr) {
            for (auto o : objs) o->loadTriangles(positions, normals, materials, count, texcoords);
        }
        void applyMaterials(MaterialSet * mat) { for (auto p : rays) p->applyMaterials(mat); }
        template<class F> void each(F f) { for (auto p : rays) f(p); }   // lights, sky, tiles: batch.each([&](psm::Pipeline * p) { ... })
        void setSeed(uint32_t seed) { master->setSeed(seed); }

        // `frames` x process(): build, camera, <= depth x (intersection, shade), sample -- `lanes` at a time
        void render(uint32_t frames, const glm::vec3 & eye, const glm::vec3 & view, uint32_t depth = 16, bool rebuild = true) {
            if (frames == 0 || ctxs.empty()) return;
            glm::mat4 persp = glm::perspective(glm::pi<float>() / 3.0f, float(master->displayWidth) / float(master->displayHeight), 0.001f, 1000.0f);
            glm::mat4 ci = glm::inverse(glm::lookAt(eye, view, glm::vec3(0.0f, 1.0f, 0.0f))), pi = glm::inverse(persp);
            float camInv[16], projInv[16];
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { camInv[4 * r + c] = ci[c][r]; projInv[4 * r + c] = pi[c][r]; }
            std::vector<uint32_t> seeds(frames);
            for (auto & sd : seeds) sd = master->nextRand();
            std::vector<psm_rt *> rts;
            std::vector<psm_bvh *> bvhs;
            size_t n = std::min<size_t>(ctxs.size(), frames);
            for (size_t s = 0; s < n; s++) { rays[s]->syncUniforms(); rts.push_back(rays[s]->handle()); bvhs.push_back(objs[s]->handle()); }
            lastResults.assign(frames, psm_lane_result{0, 0});
            check(psm_lanes_render(rts.data(), bvhs.data(), (uint32_t)n, camInv, projInv, seeds.data(), frames, depth, rebuild ? 1 : 0,
                                   nullptr, master->handle(), lastResults.data()), "FrameBatch::render");
            for (size_t s = 0; s < n; s++) rays[s]->noteTraced(objs[s]);
        }

        // The tile of a sharded frame every lane traces: the 8-row bands dealt to `world` ranks -- round-robin, or weights[r]
        // bands of every period for rank r (psm_rt_set_tile_weighted: the gathering rank also unpacks, fills and samples the
        // whole image, so bench.py gives it fewer bands) -- and the same dealing on the communicator whose gathers carry it.
        void setTile(psm_dist * dist, uint32_t rank, uint32_t world, const std::vector<uint32_t> & weights = {}) {
            const uint32_t * w = weights.empty() ? nullptr : weights.data();
            for (auto p : rays) check(psm_rt_set_tile_weighted(p->handle(), rank, world, w), "FrameBatch::setTile");
            if (dist) check(psm_dist_set_band_weights(dist, w), "FrameBatch::setTile (communicator)");
        }

        // The same frames tile-sharded over the GPUs of a node (one process per GPU; psm_dist_init has built `dist` and
        // every lane carries psm_rt_set_tile_interleaved(rank, world)): psm_dist_render_frames over all the frames -- the
        // path's one collective per frame is the tile gather to rank 0, whose accumulator() holds the image. Every rank
        // calls this with the same arguments. Set GPU_MAX_HW_QUEUES=8 in the environment before the first HIP call: the
        // runtime's default of 4 hardware queues makes lanes share queues.
        void renderSharded(psm_dist * dist, uint32_t frames, const glm::vec3 & eye, const glm::vec3 & view, uint32_t depth = 16, bool rebuild = true) {
            if (frames == 0 || ctxs.empty()) return;
            glm::mat4 persp = glm::perspective(glm::pi<float>() / 3.0f, float(master->displayWidth) / float(master->displayHeight), 0.001f, 1000.0f);
            glm::mat4 ci = glm::inverse(glm::lookAt(eye, view, glm::vec3(0.0f, 1.0f, 0.0f))), pi = glm::inverse(persp);
            float camInv[16], projInv[16];
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { camInv[4 * r + c] = ci[c][r]; projInv[4 * r + c] = pi[c][r]; }
            // all frames in one call: the lanes form two groups that alternate batches, so no batch drains before the next
            // one starts (psm_dist_render_frames); one rand() draw of the accumulating Pipeline seeds each frame
            std::vector<uint32_t> seeds(frames);
            for (auto & sd : seeds) sd = master->nextRand();
            std::vector<psm_rt *> rts;
            std::vector<psm_bvh *> bvhs;
            for (size_t s = 0; s < ctxs.size(); s++) { rays[s]->syncUniforms(); rts.push_back(rays[s]->handle()); bvhs.push_back(objs[s]->handle()); }
            check(psm_dist_render_frames(dist, rts.data(), bvhs.data(), (uint32_t)rts.size(), camInv, projInv, seeds.data(), frames, depth,
                                         rebuild ? 1 : 0, nullptr, master->handle(), nullptr), "FrameBatch::renderSharded");
            for (size_t s = 0; s < ctxs.size(); s++) rays[s]->noteTraced(objs[s]);
        }
    };
}

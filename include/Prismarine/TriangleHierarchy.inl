#include "TriangleHierarchy.hpp"

// Implementation of psm::TriangleHierarchy over the C ABI. Replaces the GL orchestration of the
// reference's Include/Prismarine/TriangleHierarchy.inl (init :49-75, allocate :77-112, loadMesh
// :173-192, build :206-329): every stage runs in libpsm_hip.so on the context's stream.

namespace NSM {

    inline TriangleHierarchy::~TriangleHierarchy() {
        if (bvh) psm_bvh_destroy(bvh);
        delete sorter;
    }

    inline void TriangleHierarchy::init() {
        sorter = new RadixSort();
    }

    // allocate(count): the reference sizes every buffer for 2*count triangles (:78)
    inline void TriangleHierarchy::allocate(const size_t &count) {
        maxt = (uint32_t)(count * 2);
        if (bvh) { psm_bvh_destroy(bvh); bvh = nullptr; }
        check(psm_bvh_create(context(), maxt, &bvh), "TriangleHierarchy::allocate");
        clearTribuffer();
    }

    inline void TriangleHierarchy::setMaterialID(int32_t id) { materialID = id; }

    inline void TriangleHierarchy::clearTribuffer() {
        markDirty();
        if (bvh) check(psm_bvh_clear(bvh), "TriangleHierarchy::clearTribuffer");
        triangleCount = 0;
    }

    inline void TriangleHierarchy::configureIntersection(bool clearDepth) { (void)clearDepth; }  // ignored by the reference's shaders too

    inline void TriangleHierarchy::loadTriangles(const float * positions, const float * normals, const int32_t * materials, size_t count) {
        if (!bvh || count == 0) return;
        int rc = psm_bvh_load_triangles(bvh, positions, normals, materials, count, materialID);
        check(rc, "TriangleHierarchy::loadTriangles");
        if (rc == PSM_OK) triangleCount += count;
        markDirty();
    }

    // loadMesh: resolve the accessor / buffer-view description on the host and append the triangles.
    // (ShadersSDK/vertex/loader.comp:32-152; geometry ingestion is row f1 of SURVEY section 8 -- it runs
    // when meshes change, not per frame.)
    inline void TriangleHierarchy::loadMesh(TriangleArrayInstance * gobject) {
        if (!gobject || gobject->meshUniformData.nodeCount <= 0 || !bvh) return;
        const MeshUniformStruct & mu = gobject->meshUniformData;
        size_t vbytes = 0, ibytes = 0;
        void * dptr = nullptr;
        if (gobject->vbo_triangle_ssbo == GLuint(-1) || psm_buf_ptr(context(), gobject->vbo_triangle_ssbo, &dptr, &vbytes) != PSM_OK) return;
        std::vector<float> iverts(vbytes / 4);
        getBufferSubData(gobject->vbo_triangle_ssbo, 0, iverts.size() * 4, iverts.data());
        std::vector<uint32_t> vindics;
        if (mu.isIndexed && gobject->vebo_triangle_ssbo != GLuint(-1) && psm_buf_ptr(context(), gobject->vebo_triangle_ssbo, &dptr, &ibytes) == PSM_OK) {
            vindics.resize(ibytes / 4);
            getBufferSubData(gobject->vebo_triangle_ssbo, 0, vindics.size() * 4, vindics.data());
        }
        auto pick = [&](uint32_t i) -> uint32_t {
            if (gobject->index16bit) return (vindics[i / 2] >> (16 * (i & 1))) & 0xFFFFu;  // vertex.glsl:199
            return vindics[i];
        };
        auto readByAccessor = [&](int accessorID, uint32_t idx, float out[4]) {  // loader.comp:32-54
            out[0] = out[1] = out[2] = out[3] = 0.f;
            if (!gobject->accessorSet || !gobject->bufferViewSet || accessorID < 0) return;
            const VirtualAccessor & ac = gobject->accessorSet->data[(size_t)accessorID];
            const VirtualBufferView & bv = gobject->bufferViewSet->data[(size_t)ac.bufferView];
            uint32_t cmps = (uint32_t)ac.components & 3u;
            uint32_t stride4 = bv.stride4 > 0 ? (uint32_t)bv.stride4 : (cmps + 1);
            size_t off = (size_t)idx * stride4 + (size_t)bv.offset4 + (size_t)ac.offset4;
            for (uint32_t k = 0; k <= cmps && off + k < iverts.size(); k++) out[k] = iverts[off + k];
        };
        glm::mat4 T = gobject->hasTransform ? gobject->transform : glm::mat4(1.0f);
        glm::mat4 Ti = glm::inverse(T);
        auto xf = [&](const glm::mat4 & m, const float v[4], float o[4]) { for (int r = 0; r < 4; r++) o[r] = m[0][r] * v[0] + m[1][r] * v[1] + m[2][r] * v[2] + m[3][r] * v[3]; };
        int mode = mu.primitiveType;
        int trp = mode == 1 ? 4 : 3;
        std::vector<float> pos, nrm;
        std::vector<int32_t> mats;
        for (int ct = 0; ct < mu.nodeCount; ct++) {
            float vertice[4][3], normal[4][3];
            for (int i = 0; i < trp; i++) {
                uint32_t ptri = (uint32_t)mu.loadingOffset + (uint32_t)(ct * trp + i);
                uint32_t vi = mu.isIndexed == 0 ? ptri : pick(ptri);
                float p[4], n[4] = {0, 0, 0, 0};
                readByAccessor(mu.vertexAccessor, vi, p);
                if (mu.normalAccessor != -1) readByAccessor(mu.normalAccessor, vi, n);
                float pv[4] = {p[0], p[1], p[2], 1.0f}, po[4];
                xf(T, pv, po);
                float nv[4] = {n[0], n[1], n[2], 0.0f}, no[4];
                // normal = vec * inverse(t): component j = dot(n, column j of inverse(t))
                for (int j = 0; j < 4; j++) no[j] = Ti[j][0] * nv[0] + Ti[j][1] * nv[1] + Ti[j][2] * nv[2] + Ti[j][3] * nv[3];
                for (int k = 0; k < 3; k++) { vertice[i][k] = po[k] / po[3]; normal[i][k] = no[k]; }
            }
            auto emit = [&](const int idx[3]) {
                float e1[3], e2[3], fn[3];
                for (int k = 0; k < 3; k++) { e1[k] = vertice[1][k] - vertice[0][k]; e2[k] = vertice[2][k] - vertice[0][k]; }
                fn[0] = e1[1] * e2[2] - e2[1] * e1[2]; fn[1] = e1[2] * e2[0] - e2[2] * e1[0]; fn[2] = e1[0] * e2[1] - e2[0] * e1[1];
                float fl = std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
                for (int k = 0; k < 3; k++) fn[k] = fl > 0 ? fn[k] / fl : 0.f;
                for (int i = 0; i < 3; i++) {
                    const float * v = vertice[idx[i]];
                    const float * n = normal[idx[i]];
                    float m = std::fmax(std::fabs(n[0]), std::fmax(std::fabs(n[1]), std::fabs(n[2])));
                    const float * use = (m >= 0.0001f && mu.normalAccessor != -1) ? n : fn;   // loader.comp:125-129
                    float l = std::sqrt(use[0] * use[0] + use[1] * use[1] + use[2] * use[2]);
                    for (int k = 0; k < 3; k++) { pos.push_back(v[k]); nrm.push_back(l > 0 ? use[k] / l : 0.f); }
                }
                mats.push_back(mu.materialID);
            };
            const int t0[3] = {0, 1, 2}, t1[3] = {3, 0, 2};   // loader.comp:57, quads -> two triangles
            emit(t0);
            if (mode == 1) emit(t1);
        }
        loadTriangles(pos.data(), nrm.data(), mats.data(), mats.size());
    }

    inline bool TriangleHierarchy::isDirty() const { return dirty; }
    inline void TriangleHierarchy::markDirty() { dirty = true; }
    inline void TriangleHierarchy::resolve() { dirty = false; }

    // build(optimization), TriangleHierarchy.inl:206-329
    inline void TriangleHierarchy::build(const glm::dmat4 &optimization) {
        if (this->triangleCount <= 0 || !dirty || !bvh) return;   // :214
        double opt[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) opt[4 * r + c] = optimization[c][r];  // glm is column-major
        check(psm_bvh_build(bvh, opt), "TriangleHierarchy::build");
        this->resolve();
    }
}

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

    inline void TriangleHierarchy::refit() { if (bvh) { check(psm_bvh_refit(bvh), "TriangleHierarchy::refit"); this->resolve(); } }
    inline void TriangleHierarchy::setBuildGraph(bool enable) { if (bvh) check(psm_bvh_set_build_graph(bvh, enable ? 1 : 0), "TriangleHierarchy::setBuildGraph"); }
    inline void TriangleHierarchy::configureIntersection(bool clearDepth) { (void)clearDepth; }  // ignored by the reference's shaders too

    inline void TriangleHierarchy::loadTriangles(const float * positions, const float * normals, const int32_t * materials, size_t count, const float * texcoords) {
        if (!bvh || count == 0) return;
        int rc = psm_bvh_load_triangles(bvh, positions, normals, materials, count, materialID);
        check(rc, "TriangleHierarchy::loadTriangles");
        if (rc == PSM_OK && texcoords) check(psm_bvh_set_texcoords(bvh, triangleCount, texcoords, count), "TriangleHierarchy::loadTriangles(texcoords)");
        if (rc == PSM_OK) triangleCount += count;
        markDirty();
    }

    // loadMesh(gobject), TriangleHierarchy.inl:173-192: the accessor / buffer-view description is resolved by
    // the HIP gather kernel behind psm_bvh_load_mesh (vertex/loader.comp:32-152); no host round trip.
    inline void TriangleHierarchy::loadMesh(TriangleArrayInstance * gobject) {
        if (!gobject || gobject->meshUniformData.nodeCount <= 0 || !bvh) return;
        if (!gobject->accessorSet || !gobject->bufferViewSet) return;
        const MeshUniformStruct & mu = gobject->meshUniformData;
        psm_mesh_desc d;
        std::memset(&d, 0, sizeof(d));
        void * vptr = nullptr; size_t vbytes = 0, ibytes = 0;
        if (gobject->vbo_triangle_ssbo == GLuint(-1) || psm_buf_ptr(context(), gobject->vbo_triangle_ssbo, &vptr, &vbytes) != PSM_OK) return;
        d.d_vertices = (const float *)vptr; d.vertex_floats = vbytes / 4;
        if (mu.isIndexed && gobject->vebo_triangle_ssbo != GLuint(-1)) {
            void * iptr = nullptr;
            if (psm_buf_ptr(context(), gobject->vebo_triangle_ssbo, &iptr, &ibytes) == PSM_OK) { d.d_indices = (const uint32_t *)iptr; d.index_words = ibytes / 4; }
        }
        std::vector<psm_accessor> acc;
        for (const VirtualAccessor & a : gobject->accessorSet->data) acc.push_back(psm_accessor{a.offset4, (int32_t)(a.components & 3), a.bufferView});
        std::vector<psm_buffer_view> views;
        for (const VirtualBufferView & v : gobject->bufferViewSet->data) views.push_back(psm_buffer_view{v.offset4, v.stride4});
        d.accessors = acc.data(); d.accessor_count = (uint32_t)acc.size();
        d.views = views.data(); d.view_count = (uint32_t)views.size();
        d.vertex_accessor = mu.vertexAccessor; d.normal_accessor = mu.normalAccessor;
        d.texcoord_accessor = mu.texcoordAccessor; d.modifier_accessor = mu.modifierAccessor;
        glm::mat4 T = gobject->hasTransform ? gobject->transform : glm::mat4(1.0f);
        glm::mat4 Ti = glm::inverse(T);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { d.transform[4 * r + c] = T[c][r]; d.transform_inv[4 * r + c] = Ti[c][r]; }
        d.material_id = mu.materialID; d.is_indexed = mu.isIndexed; d.index16 = gobject->index16bit ? 1 : 0;
        d.node_count = mu.nodeCount; d.primitive_type = mu.primitiveType; d.loading_offset = mu.loadingOffset;
        int rc = psm_bvh_load_mesh(bvh, &d);
        check(rc, "TriangleHierarchy::loadMesh");
        if (rc == PSM_OK) triangleCount += (size_t)mu.nodeCount * (mu.primitiveType == 1 ? 2 : 1);
        markDirty();
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

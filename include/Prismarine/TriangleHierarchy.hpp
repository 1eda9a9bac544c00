#pragma once
// Prismarine/TriangleHierarchy.hpp -- psm::TriangleHierarchy, same public surface as the reference
// (Include/Prismarine/TriangleHierarchy.hpp:75-94): geometry store + HLBVH build.

#include "Utils.hpp"
#include "VertexInstance.hpp"
#include "Radix.hpp"

namespace NSM {

    class TriangleHierarchy : public BaseClass {
    protected:
        friend class Pipeline;
        RadixSort * sorter = nullptr;
        bool dirty = false;
        uint32_t maxt = 1024 * 128 * 1;
        psm_bvh * bvh = nullptr;
        void init();

    public:
        TriangleHierarchy() { init(); }
        ~TriangleHierarchy();

        int32_t materialID = 0;
        size_t triangleCount = 0;

        void syncUniforms() {}
        void allocate(const size_t &count);
        void setMaterialID(int32_t id);
        void bindUniforms() {}
        void bind() {}
        void bindBVH() {}
        void bindLeafs() {}
        void clearTribuffer();
        void loadMesh(TriangleArrayInstance * gobject);
        // direct ingestion of a world-space triangle soup (9 floats per triangle; normals / material ids optional)
        void loadTriangles(const float * positions, const float * normals, const int32_t * materials, size_t count, const float * texcoords = nullptr);
        bool isDirty() const;
        void markDirty();
        void resolve();
        void build(const glm::dmat4 &optimization = glm::dmat4(1.0));
        void configureIntersection(bool clearDepth);
        void refit();                     // not in the reference: boxes only, for triangles reloaded in place; the tree is the last build's
        void setBuildGraph(bool enable);  // not in the reference: replay rebuilds as one captured hipGraph (default on)
        psm_bvh * handle() { return bvh; }
    };
}

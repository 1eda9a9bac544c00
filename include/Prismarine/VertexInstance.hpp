#pragma once
// Prismarine/VertexInstance.hpp -- accessor / buffer-view data holders and TriangleArrayInstance
// (reference Include/Prismarine/VertexInstance.{hpp,inl}). Pure host-side description of a mesh;
// TriangleHierarchy::loadMesh resolves it (the job of ShadersSDK/vertex/loader.comp).

#include "Utils.hpp"
#include "Structs.hpp"

namespace NSM {

    class VertexInstance;

    template<int BINDING, class STRUCTURE>
    class BufferComposer : public BaseClass {
    public:
        BufferComposer() {}
        friend VertexInstance;
        friend class TriangleHierarchy;
        int32_t addElement(STRUCTURE accessorDesc) { int32_t ptr = (int32_t)data.size(); data.push_back(accessorDesc); return ptr; }
        void bind() {}
    protected:
        GLuint buffer = GLuint(-1);
        std::vector<STRUCTURE> data;
    };

    using AccessorSet = BufferComposer<7, VirtualAccessor>;
    using BufferViewSet = BufferComposer<8, VirtualBufferView>;

    class VertexInstance : public BaseClass {};

    class TriangleArrayInstance : public VertexInstance {
    public:
        TriangleArrayInstance() {}
        friend class TriangleHierarchy;

        size_t getNodeCount() { return (size_t)meshUniformData.nodeCount; }
        void setNodeCount(size_t tcount) { meshUniformData.nodeCount = (GLint)tcount; }
        void setMaterialOffset(int32_t id) { meshUniformData.materialID = id; }
        void useIndex16bit(bool b16) { index16bit = b16; }
        void setTransform(glm::mat4 t) { meshUniformData.transform = glm::transpose(t); meshUniformData.transformInv = glm::inverse(t); hasTransform = true; transform = t; }
        void setTransform(glm::dmat4 t) { this->setTransform(glm::mat4(t)); }
        void setIndexed(const int32_t b) { meshUniformData.isIndexed = b; }
        void setVertices(const GLuint &buf) { vbo_triangle_ssbo = buf; }
        void setIndices(const GLuint &buf, const bool &all = true) { (void)all; vebo_triangle_ssbo = buf; }
        void setLoadingOffset(const int32_t &off) { meshUniformData.loadingOffset = off; }
        void bind() {}

        void setVertexAccessor(int32_t accessorID) { meshUniformData.vertexAccessor = accessorID; }
        void setNormalAccessor(int32_t accessorID) { meshUniformData.normalAccessor = accessorID; }
        void setTexcoordAccessor(int32_t accessorID) { meshUniformData.texcoordAccessor = accessorID; }
        void setModifierAccessor(int32_t accessorID) { meshUniformData.modifierAccessor = accessorID; }
        void setAccessorSet(AccessorSet * accessorSet) { this->accessorSet = accessorSet; }
        void setBufferViewSet(BufferViewSet * bufferViewSet) { this->bufferViewSet = bufferViewSet; }

    protected:
        bool index16bit = false;
        bool hasTransform = false;
        glm::mat4 transform;
        GLuint vbo_triangle_ssbo = GLuint(-1);
        GLuint vebo_triangle_ssbo = GLuint(-1);
        BufferViewSet * bufferViewSet = nullptr;
        AccessorSet * accessorSet = nullptr;
        MeshUniformStruct meshUniformData;
    };
}

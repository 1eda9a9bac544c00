#pragma once
// Prismarine/MaterialSet.hpp -- psm::MaterialSet (reference MaterialSet.{hpp,inl}): a host array of
// VirtualMaterial handed to Pipeline::applyMaterials.

#include "Utils.hpp"
#include "Structs.hpp"
#include "TextureSet.hpp"

namespace NSM {
    class TriangleHierarchy;
    class Pipeline;

    class MaterialSet : public BaseClass {
    protected:
        friend class Pipeline;
        friend class TriangleHierarchy;
        TextureSet * texset = nullptr;
        std::vector<VirtualMaterial> submats;
        GLint loadOffset = 0;
        uint64_t revision = 1;

    public:
        MaterialSet() {}

        void setTextureSet(TextureSet *txs) { texset = txs; }
        void setTextureSet(TextureSet &txs) { texset = &txs; }
        void clearSubmats() { submats.resize(0); revision++; }

        size_t getMaterialCount() { return submats.size(); }
        size_t addSubmat(const VirtualMaterial * submat) { size_t idx = submats.size(); submats.push_back(*submat); revision++; return idx; }
        size_t addSubmat(const VirtualMaterial &submat) { return this->addSubmat(&submat); }
        void setSumbat(const size_t& i, const VirtualMaterial &submat) { if (submats.size() <= i) submats.resize(i + 1); submats[i] = submat; revision++; }

        void loadToVGA() { if (texset) texset->loadToVGA(); }   // the upload happens in Pipeline::applyMaterials
        void bindWithContext(GLuint & prog) { (void)prog; }
        void setLoadingOffset(GLint loadOffset) { this->loadOffset = loadOffset; revision++; }
    };
}

#pragma once
// Prismarine/TextureSet.hpp -- texture table holder (reference TextureSet.{hpp,inl}). Material
// textures are a later row: the table is kept so application code compiles; the
// texture-less material path is what the kernels implement.

#include "Utils.hpp"
#include "Structs.hpp"

namespace NSM {
    class TextureSet : public BaseClass {
    public:
        TextureSet() {}
        void freeTexture(const uint32_t& idx) { if (idx < textures.size()) textures[idx] = GLuint(-1); }
        void clearGlTextures() { textures.clear(); }
        void setTexture(GLuint location, const GLuint & texnum) { if (textures.size() <= location) textures.resize(location + 1, GLuint(-1)); textures[location] = texnum; }
        uint32_t loadTexture(const GLuint & gltexture) { textures.push_back(gltexture); return (uint32_t)textures.size(); }
        void loadToVGA() {}
        void bindWithContext(GLuint & prog) { (void)prog; }
    private:
        std::vector<GLuint> textures;
    };
}

#pragma once
// Prismarine/TextureSet.hpp -- psm::TextureSet (reference TextureSet.{hpp,inl}): the table of material
// textures that surface.comp indexes with diffusePart / specularPart / bumpPart / emissivePart.
// GL-free: a "texture name" is an RGBA8 image registered with psm::createTextureRGBA8() (what
// glCreateTextures + glTextureSubImage2D were, TextureSet.inl:111-118); slot numbering, slot 0 = none and
// the free list follow TextureSet.inl:7-12,42-86. Pipeline::applyMaterials uploads changed slots through
// psm_rt_set_texture (at most 31 usable slots, MAX_TEXTURES = 32, surface.comp:46).

#include <map>
#include <string>
#include <vector>
#include "Utils.hpp"
#include "Structs.hpp"

namespace NSM {
    struct HostTexture { std::vector<uint8_t> rgba8; uint32_t width = 0, height = 0; };
    inline std::vector<HostTexture> & hostTextures() { static std::vector<HostTexture> t(1); return t; }
    // rows bottom-up as GL stores them: row 0 is v = 0
    inline GLuint createTextureRGBA8(const uint8_t * rgba8, uint32_t width, uint32_t height) {
        HostTexture h; h.width = width; h.height = height;
        h.rgba8.assign(rgba8, rgba8 + (size_t)width * height * 4);
        hostTextures().push_back(std::move(h));
        return (GLuint)(hostTextures().size() - 1);
    }

    class Pipeline;

    class TextureSet : public BaseClass {
    protected:
        friend class Pipeline;
        uint64_t revision = 1;
        void init() { textures.assign(1, GLuint(-1)); freedomTextures.clear(); texnames.clear(); }

    public:
        std::vector<uint32_t> textures;
        std::vector<uint32_t> freedomTextures;
        std::map<std::string, uint32_t> texnames;

        TextureSet() { init(); }
        void loadToVGA() {}                                    // the upload happens in Pipeline::applyMaterials
        void bindWithContext(GLuint & prog) { (void)prog; }

        void freeTexture(const uint32_t & idx) { if (idx < textures.size()) { freedomTextures.push_back(idx); textures[idx] = GLuint(-1); revision++; } }
        void freeTextureByGL(const GLuint & gltexture) { for (size_t i = 1; i < textures.size(); i++) if (textures[i] == gltexture) freeTexture((uint32_t)i); }
        void clearGlTextures() { for (size_t i = 1; i < textures.size(); i++) freeTexture((uint32_t)i); }
        uint32_t getTexture(const GLuint & gltexture) { for (size_t i = 1; i < textures.size(); i++) if (textures[i] == gltexture && textures[i] != GLuint(-1)) return (uint32_t)i; return 0; }
        GLuint getGLTexture(const uint32_t & idx) { return textures[idx]; }

        uint32_t loadTexture(const GLuint & gltexture) {       // TextureSet.inl:73-86
            uint32_t idx = getTexture(gltexture);
            if (idx) return idx;
            if (!freedomTextures.empty()) { idx = freedomTextures.back(); freedomTextures.pop_back(); textures[idx] = gltexture; }
            else { idx = (uint32_t)textures.size(); textures.push_back(gltexture); }
            revision++;
            return idx;
        }
        // the FreeImage overload (TextureSet.inl:90-122) decoded a file; here the caller hands the pixels
        uint32_t loadTexture(const std::string & name, const uint8_t * rgba8, uint32_t width, uint32_t height, bool force_write = false) {
            if (name == "" || !rgba8) return 0;
            if (!force_write && texnames.find(name) != texnames.end()) return getTexture(texnames[name]);
            GLuint t = createTextureRGBA8(rgba8, width, height);
            texnames[name] = t;
            return loadTexture(t);
        }
    };
}

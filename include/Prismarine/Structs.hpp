#pragma once
// Prismarine/Structs.hpp -- host mirrors of the device structs (layout contract of the reference's
// Include/Prismarine/Structs.hpp:86-262 and ShadersSDK/include/structs.glsl:17-204). The kernels keep
// rays / hits in SoA form internally; these AoS types remain for API compatibility and for the
// snap / debug read-backs.

#include "Utils.hpp"

namespace NSM {

    typedef float Vc1;
    typedef int32_t iVc1;
    struct Vc2 { float x, y; };
    struct Vc3 { float x, y, z; };
    struct Vc4 { float x, y, z, w; };
    struct Vc4x4 { Vc4 m0, m1, m2, m3; };
    struct iVc2 { int32_t x, y; };
    struct iVc3 { int32_t x, y, z; };
    struct iVc4 { int32_t x, y, z, w; };

    struct bbox { glm::vec4 mn; glm::vec4 mx; };

    struct Ray {                       // RayRework, 80 bytes
        glm::vec4 origin, direct, color, final;
        int bitfield, idx, texel, hit;
    };
    struct Hit {                       // HitRework, 112 bytes
        glm::vec4 uvt, normalHeight, tangent, texcoord;
        glm::uvec4 metallicRoughness, emission_albedo;
        int bitfield, ray, materialID, next;
    };
    struct Texel { Vc4 coord, last3d; iVc4 EXT; };
    struct HlbvhNode { glm::uvec4 box; iVc4 pdata; };   // 32 bytes: fp16 box + (children | leaf, parent, triangle)
    struct ColorChain { Vc4 color = {0, 0, 0, 0}; iVc4 cdata = {0, 0, 0, 0}; };

    typedef psm_light LightUniformStruct;               // Structs.hpp:165-170
    typedef psm_material VirtualMaterial;               // Structs.hpp:240-262 (128 bytes)

    struct MeshUniformStruct {                          // Structs.hpp:207-225
        GLint vertexAccessor = -1, normalAccessor = -1, texcoordAccessor = -1, modifierAccessor = -1;
        glm::mat4 transform, transformInv;
        GLint materialID = 0, isIndexed = 0, nodeCount = 1, primitiveType = 0;
        GLint loadingOffset = 0, storingOffset = 0, _reserved0 = 1, _reserved1 = 2;
    };
    struct VirtualBufferView { GLint offset4 = 0; GLint stride4 = 1; };
    struct VirtualAccessor { GLint offset4 = 0; GLint components : 2, type : 4, normalized : 1; GLint bufferView = -1; };

    inline VirtualMaterial makeMaterial() {
        VirtualMaterial m;
        std::memset(&m, 0, sizeof(m));
        m.ior = 1.0f;
        m.roughness = 0.0001f;
        return m;
    }

    static_assert(sizeof(Ray) == 80, "Ray layout");
    static_assert(sizeof(Hit) == 112, "Hit layout");
    static_assert(sizeof(HlbvhNode) == 32, "HlbvhNode layout");
    static_assert(sizeof(VirtualMaterial) == 128, "VirtualMaterial layout");
}

// Drop-in check for scene loading: the reference viewer's glTF loop (Source/Examples/Viewer.cpp:133-277) written against
// include/Prismarine -- raw buffers, BufferViewSet, one TriangleArrayInstance per primitive with its AccessorSet, the node walk in
// glm doubles, setTransform + TriangleHierarchy::loadMesh per (node, primitive) -- with the parsed file handed over in a flat
// binary form (tests/test_gpu_parity.py writes it from the .gltf's JSON; the reference uses tinygltf for that step).
// usage: gltf_load <model.bin> <out.bin>      out: int32 n, then n*9 positions, n*9 normals, n*6 texcoords, n material ids
// model.bin (little endian):
//   int32 nbuffers   { int32 bytes; bytes... (padded to 4) }
//   int32 nviews     { int32 buffer, byteOffset, byteStride }
//   int32 naccessors { int32 bufferView, byteOffset, componentType, count }
//   int32 nmeshes    { int32 nprims { int32 position, normal, texcoord, indices, material, mode } }     (accessor ids, -1 = none)
//   int32 nnodes     { int32 mesh; int32 hasMatrix, hasT, hasS, hasR; double matrix[16], T[3], S[3], R[4]; int32 nchildren; int32 children... }
//   int32 nroots     { int32 node }
//   double mscale
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "Prismarine/Prismarine.hpp"
#include "Prismarine/Implementations.hpp"

struct View { int32_t buffer, byteOffset, byteStride; };
struct Acc { int32_t bufferView, byteOffset, componentType, count; };
struct Prim { std::map<std::string, int> attributes; int32_t indices, material, mode; };
struct Node { int32_t mesh; std::vector<double> matrix, translation, scale, rotation; std::vector<int> children; };

static FILE * in = nullptr;
template <class T> static T rd() { T v; if (std::fread(&v, sizeof(T), 1, in) != 1) { std::fprintf(stderr, "short model file\n"); std::exit(2); } return v; }

int main(int argc, char ** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage\n"); return 2; }
    in = std::fopen(argv[1], "rb");
    if (!in) return 2;
    std::vector<std::vector<uint8_t>> buffers((size_t)rd<int32_t>());
    for (auto & b : buffers) { b.resize((size_t)rd<int32_t>()); if (std::fread(b.data(), 1, b.size(), in) != b.size()) return 2; }
    std::vector<View> bufferViews((size_t)rd<int32_t>());
    for (auto & v : bufferViews) v = rd<View>();
    std::vector<Acc> accessors((size_t)rd<int32_t>());
    for (auto & a : accessors) a = rd<Acc>();
    std::vector<std::vector<Prim>> meshes((size_t)rd<int32_t>());
    for (auto & m : meshes) {
        m.resize((size_t)rd<int32_t>());
        for (auto & p : m) {
            int32_t pos = rd<int32_t>(), nor = rd<int32_t>(), tex = rd<int32_t>();
            if (pos >= 0) p.attributes["POSITION"] = pos;
            if (nor >= 0) p.attributes["NORMAL"] = nor;
            if (tex >= 0) p.attributes["TEXCOORD_0"] = tex;
            p.indices = rd<int32_t>(); p.material = rd<int32_t>(); p.mode = rd<int32_t>();
        }
    }
    std::vector<Node> nodes((size_t)rd<int32_t>());
    for (auto & n : nodes) {
        n.mesh = rd<int32_t>();
        int32_t has[4]; for (int k = 0; k < 4; k++) has[k] = rd<int32_t>();
        double m[16], t[3], s[3], r[4];
        for (double & v : m) v = rd<double>();
        for (double & v : t) v = rd<double>();
        for (double & v : s) v = rd<double>();
        for (double & v : r) v = rd<double>();
        if (has[0]) n.matrix.assign(m, m + 16);
        if (has[1]) n.translation.assign(t, t + 3);
        if (has[2]) n.scale.assign(s, s + 3);
        if (has[3]) n.rotation.assign(r, r + 4);
        n.children.resize((size_t)rd<int32_t>());
        for (int & c : n.children) c = rd<int32_t>();
    }
    std::vector<int> roots((size_t)rd<int32_t>());
    for (int & r : roots) r = rd<int32_t>();
    double mscale = rd<double>();
    std::fclose(in);

    // ---- GltfViewer::init from here on -------------------------------------------------------------------------------------
    // make raw mesh buffers, Viewer.cpp:133-139 (glCreateBuffers + glNamedBufferData)
    std::vector<GLuint> glBuffers;
    for (auto & b : buffers) {
        GLuint glBuf = psm::allocateBuffer<uint8_t>(b.size());
        psm::bufferSubData(glBuf, 0, b.size(), b.data());
        glBuffers.push_back(glBuf);
    }
    // make buffer views, :141-148
    psm::BufferViewSet * bfvi = new psm::BufferViewSet();
    for (auto const & bv : bufferViews) {
        psm::VirtualBufferView bfv;
        bfv.offset4 = bv.byteOffset / 4;
        bfv.stride4 = bv.byteStride / 4;
        bfvi->addElement(bfv);
    }
    // load mesh templates, :151-232
    std::vector<std::vector<psm::TriangleArrayInstance *>> meshVec;
    for (size_t m = 0; m < meshes.size(); m++) {
        std::vector<psm::TriangleArrayInstance *> primitiveVec;
        for (size_t i = 0; i < meshes[m].size(); i++) {
            Prim & prim = meshes[m][i];
            psm::TriangleArrayInstance * geom = new psm::TriangleArrayInstance();
            psm::AccessorSet * acs = new psm::AccessorSet();
            geom->setAccessorSet(acs);
            geom->setBufferViewSet(bfvi);
            for (auto const & it : prim.attributes) {
                Acc & accessor = accessors[(size_t)it.second];
                View & bufferView = bufferViews[(size_t)accessor.bufferView];
                psm::VirtualAccessor vattr;
                vattr.offset4 = accessor.byteOffset / 4;
                vattr.bufferView = accessor.bufferView;
                if (it.first.compare("POSITION") == 0) {
                    vattr.components = 3 - 1;
                    geom->setVertices(glBuffers[(size_t)bufferView.buffer]);
                    geom->setVertexAccessor(acs->addElement(vattr));
                } else if (it.first.compare("NORMAL") == 0) {
                    vattr.components = 3 - 1;
                    geom->setNormalAccessor(acs->addElement(vattr));
                } else if (it.first.compare("TEXCOORD_0") == 0) {
                    vattr.components = 2 - 1;
                    geom->setTexcoordAccessor(acs->addElement(vattr));
                }
            }
            if (prim.indices >= 0) {
                Acc & idcAccessor = accessors[(size_t)prim.indices];
                View & bufferView = bufferViews[(size_t)idcAccessor.bufferView];
                geom->setNodeCount((size_t)idcAccessor.count / 3);
                geom->setIndices(glBuffers[(size_t)bufferView.buffer]);
                bool isInt16 = idcAccessor.componentType == 5122 || idcAccessor.componentType == 5123;
                int32_t loadingOffset = (bufferView.byteOffset + idcAccessor.byteOffset) / (isInt16 ? 2 : 4);
                geom->setLoadingOffset(loadingOffset);
                geom->setIndexed(true);
                geom->useIndex16bit(isInt16);
            }
            geom->setMaterialOffset(prim.material);
            if (prim.mode == 4) primitiveVec.push_back(geom);
        }
        meshVec.push_back(primitiveVec);
    }
    // create geometry intersector, :235-242
    psm::TriangleHierarchy * intersector = new psm::TriangleHierarchy();
    intersector->allocate(1024 * 16);
    glm::dmat4 matrix(1.0);
    matrix *= glm::scale(glm::dvec3(mscale));
    intersector->clearTribuffer();
    // load meshes, :244-277
    std::function<void(Node &, glm::dmat4, int)> traverse = [&](Node & node, glm::dmat4 inTransform, int recursive) -> void {
        glm::dmat4 localTransform(1.0);
        localTransform *= (node.matrix.size() >= 16 ? glm::make_mat4(node.matrix.data()) : glm::dmat4(1.0));
        localTransform *= (node.translation.size() >= 3 ? glm::translate(glm::make_vec3(node.translation.data())) : glm::dmat4(1.0));
        localTransform *= (node.scale.size() >= 3 ? glm::scale(glm::make_vec3(node.scale.data())) : glm::dmat4(1.0));
        localTransform *= (node.rotation.size() >= 4 ? glm::mat4_cast(glm::make_quat(node.rotation.data())) : glm::dmat4(1.0));
        glm::dmat4 transform = inTransform * localTransform;
        if (node.mesh >= 0) {
            std::vector<psm::TriangleArrayInstance *> & mesh = meshVec[(size_t)node.mesh];
            for (size_t p = 0; p < mesh.size(); p++) {
                psm::TriangleArrayInstance * geom = mesh[p];
                geom->setTransform(transform);
                intersector->loadMesh(geom);
            }
        } else if (node.children.size() > 0) {
            for (size_t n = 0; n < node.children.size(); n++) {
                if (recursive >= 0) traverse(nodes[(size_t)node.children[n]], transform, recursive - 1);
            }
        }
    };
    for (size_t n = 0; n < roots.size(); n++) traverse(nodes[(size_t)roots[n]], glm::dmat4(matrix), 2);

    // ---- what the hierarchy holds now ----------------------------------------------------------------------------------------
    int32_t n = (int32_t)intersector->triangleCount;
    std::vector<float> pos((size_t)n * 9), nrm((size_t)n * 9), tex((size_t)n * 6);
    std::vector<int32_t> mats((size_t)n);
    int rc = psm_bvh_download(intersector->handle(), PSM_BVH_POSITIONS, pos.data(), pos.size() * 4);
    rc |= psm_bvh_download(intersector->handle(), PSM_BVH_NORMALS, nrm.data(), nrm.size() * 4);
    rc |= psm_bvh_download(intersector->handle(), PSM_BVH_TEXCOORDS, tex.data(), tex.size() * 4);
    rc |= psm_bvh_download(intersector->handle(), PSM_BVH_MATERIALS, mats.data(), mats.size() * 4);
    if (rc != PSM_OK) { std::fprintf(stderr, "download failed: %s\n", psm_last_error(psm::context())); return 1; }
    FILE * o = std::fopen(argv[2], "wb");
    std::fwrite(&n, 4, 1, o);
    std::fwrite(pos.data(), 4, pos.size(), o);
    std::fwrite(nrm.data(), 4, nrm.size(), o);
    std::fwrite(tex.data(), 4, tex.size(), o);
    std::fwrite(mats.data(), 4, mats.size(), o);
    std::fclose(o);
    std::printf("triangles %d\n", n);
    delete intersector;
    return 0;
}

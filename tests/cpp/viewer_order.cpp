// Drop-in check: the call sequence of the reference's GltfViewer::init/process
// (Source/Examples/Viewer.cpp:56-63, 231-242, 296-312) written against include/Prismarine.
// usage: viewer_order <triangles.bin> <w> <h> <frames> <out.bin> [lanes]
//   lanes given: the same frames through psm::FrameBatch (several frames in flight, FrameBatch.hpp)
//   triangles.bin: int32 n, then n*9 float positions, n*9 float normals, n int32 material ids,
//                  int32 m, then m * (4 float diffuse, 4 float specular)
#include <cstdio>
#include <vector>

#include "Prismarine/Prismarine.hpp"
#include "Prismarine/Implementations.hpp"

int main(int argc, char ** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage\n"); return 2; }
    FILE * f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t n = 0, m = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    std::vector<float> pos((size_t)n * 9), nrm((size_t)n * 9);
    std::vector<int32_t> mats((size_t)n);
    if (std::fread(pos.data(), 4, pos.size(), f) != pos.size() || std::fread(nrm.data(), 4, nrm.size(), f) != nrm.size() ||
        std::fread(mats.data(), 4, mats.size(), f) != mats.size() || std::fread(&m, 4, 1, f) != 1) return 2;
    std::vector<float> md((size_t)m * 8);
    if (std::fread(md.data(), 4, md.size(), f) != md.size()) return 2;
    float eye[3], view[3];
    if (std::fread(eye, 4, 3, f) != 3 || std::fread(view, 4, 3, f) != 3) return 2;
    std::fclose(f);
    uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]);
    int frames = std::atoi(argv[4]);

    int lanes = argc > 6 ? std::atoi(argv[6]) : 0;

    // GltfViewer::init
    psm::MaterialSet * materialManager = new psm::MaterialSet();
    for (int i = 0; i < m; i++) {
        psm::VirtualMaterial vm = psm::makeMaterial();
        for (int k = 0; k < 4; k++) { vm.diffuse[k] = md[8 * i + k]; vm.specular[k] = md[8 * i + 4 + k]; }
        materialManager->addSubmat(vm);
    }
    if (lanes > 0) {   // frames x process() with `lanes` of them in flight
        psm::FrameBatch batch((uint32_t)lanes, w, h);
        batch.setSeed(31337);
        batch.allocate((size_t)n);
        batch.clearTribuffer();
        batch.loadTriangles(pos.data(), nrm.data(), mats.data(), (size_t)n);
        batch.applyMaterials(materialManager);
        batch.render((uint32_t)frames, glm::vec3(eye[0], eye[1], eye[2]), glm::vec3(view[0], view[1], view[2]), 16);
        psm::Pipeline::HdrImage bi = batch.accumulator()->snapHdr();
        FILE * bo = std::fopen(argv[5], "wb");
        std::fwrite(bi.image, 4, (size_t)bi.width * bi.height * 4, bo);
        std::fclose(bo);
        delete[] bi.image;
        uint64_t traced = 0;
        for (auto & r : batch.lastResults) traced += r.rays;
        std::printf("frames %d lanes %d rays %llu\n", frames, lanes, (unsigned long long)traced);
        delete materialManager;
        return 0;
    }
    psm::Pipeline * rays = new psm::Pipeline();
    rays->setSeed(31337);
    rays->resizeBuffers(w, h);
    rays->resize(w, h);
    psm::TriangleHierarchy * intersector = new psm::TriangleHierarchy();
    intersector->allocate((size_t)n);
    intersector->clearTribuffer();
    intersector->loadTriangles(pos.data(), nrm.data(), mats.data(), (size_t)n);

    const int32_t depth = 16;
    for (int fr = 0; fr < frames; fr++) {
        // GltfViewer::process, Viewer.cpp:296-312
        materialManager->loadToVGA();
        intersector->markDirty();
        intersector->build();
        rays->camera(glm::vec3(eye[0], eye[1], eye[2]), glm::vec3(view[0], view[1], view[2]));
        for (int32_t j = 0; j < depth; j++) {
            if (rays->getRayCount() <= 0) break;
            rays->intersection(intersector);
            rays->applyMaterials(materialManager);
            rays->shade();
            rays->reclaim();
        }
        rays->sample();
        rays->render();
    }
    psm::Pipeline::HdrImage img = rays->snapHdr();
    FILE * o = std::fopen(argv[5], "wb");
    std::fwrite(img.image, 4, (size_t)img.width * img.height * 4, o);
    std::fclose(o);
    delete[] img.image;
    delete rays; delete intersector; delete materialManager;
    return 0;
}

// glm_pin.cpp -- TEST INFRASTRUCTURE (oracle/): the two host-side formulas of the hot path evaluated with the
// glm the reference vendors (/root/reference/External/include/glm, included where it lies -- nothing is copied),
// through exactly the glm calls the reference makes:
//   fit transform    Include/Prismarine/TriangleHierarchy.inl:257-267
//   camera matrices  Include/Prismarine/Pipeline.inl:279-312 (lookAt up = +Y, perspective(pi/3, aspect, 0.001, 1000))
//   glTF node transforms  Source/Examples/Viewer.cpp:240-253 (root scale, matrix * T * S * R under the parent's) and what
//                    TriangleArrayInstance::setTransform makes of them (Include/Prismarine/VertexInstance.inl:54-58)
// Built into oracle/_ref/libglm_pin.so by `make -C oracle ref` (only where /root/reference exists); used by
// tests/golden/make_glm_golden.py to pin the oracle's / the product's own restatements of these formulas.
#define GLM_ENABLE_EXPERIMENTAL
#include "glm/glm.hpp"
#include "glm/gtc/matrix_transform.hpp"
#include "glm/gtc/quaternion.hpp"
#include "glm/gtc/type_ptr.hpp"
#include "glm/gtx/transform.hpp"

#include <cstring>

extern "C" {

// mn, mx: the reduced bounds (4 floats each, as bbox.mn / bbox.mx); opt: the dmat4 `optimization` in glm's
// column-major memory order. Outputs: geometryUniformData.transform / transformInv exactly as uploaded
// (value_ptr of the transposed float matrix: 16 floats each).
void glm_pin_fit(const float* mn, const float* mx, const double* opt, float* transform, float* transformInv) {
    glm::vec3 scale = glm::vec3(glm::make_vec4(mx) - glm::make_vec4(mn));
    glm::vec3 offset = glm::vec3(glm::make_vec4(mn));
    glm::dmat4 optimization = glm::make_mat4(opt);
    glm::dmat4 mat(1.0);
    mat *= glm::inverse(glm::translate(glm::dvec3(offset)) * glm::scale(glm::dvec3(scale)));
    mat *= glm::inverse(glm::dmat4(optimization));
    glm::mat4 t = glm::transpose(glm::mat4(mat));
    glm::mat4 ti = glm::transpose(glm::inverse(glm::mat4(mat)));
    std::memcpy(transform, glm::value_ptr(t), 64);
    std::memcpy(transformInv, glm::value_ptr(ti), 64);
}

// cameraUniformData.camInv / projInv exactly as uploaded (Pipeline.inl:283-284) for camera(eye, view) at a display
// of width x height (Pipeline.inl:298-312, default up vector)
void glm_pin_camera(const float* eye, const float* view, int width, int height, float* camInv, float* projInv) {
    glm::mat4 persp = glm::perspective(glm::pi<float>() / 3.0f, float(width) / float(height), 0.001f, 1000.0f);
    glm::mat4 sidemat = glm::lookAt(glm::make_vec3(eye), glm::make_vec3(view), glm::vec3(0.0f, 1.0f, 0.0f));
    glm::mat4 ci = glm::transpose(glm::inverse(sidemat));
    glm::mat4 pi = glm::transpose(glm::inverse(persp));
    std::memcpy(camInv, glm::value_ptr(ci), 64);
    std::memcpy(projInv, glm::value_ptr(pi), 64);
}

// inverse(dmat4(optimization)) cast to float and transposed: the first-pass transform of minmax.comp
// (TriangleHierarchy.inl:226-232)
void glm_pin_inverse_opt(const double* opt, float* transform) {
    glm::dmat4 mat(1.0);
    mat *= glm::inverse(glm::dmat4(glm::make_mat4(opt)));
    glm::mat4 t = glm::transpose(glm::mat4(mat));
    std::memcpy(transform, glm::value_ptr(t), 64);
}

// Viewer.cpp:240-241: glm::dmat4 matrix(1.0); matrix *= glm::scale(glm::dvec3(mscale));  (glm memory order, 16 doubles)
void glm_pin_gltf_root(double mscale, double* out) {
    glm::dmat4 matrix(1.0);
    matrix *= glm::scale(glm::dvec3(mscale));
    std::memcpy(out, glm::value_ptr(matrix), 128);
}

// Viewer.cpp:246-253: the transform of a node whose parent's is `parent`. A property the node does not have is passed as
// NULL (tinygltf leaves its vector empty: Viewer.cpp then multiplies by the identity).
void glm_pin_gltf_node(const double* parent, const double* matrix, const double* translation, const double* scale,
                       const double* rotation, double* out) {
    glm::dmat4 inTransform = glm::make_mat4(parent);
    glm::dmat4 localTransform(1.0);
    localTransform *= (matrix ? glm::make_mat4(matrix) : glm::dmat4(1.0));
    localTransform *= (translation ? glm::translate(glm::make_vec3(translation)) : glm::dmat4(1.0));
    localTransform *= (scale ? glm::scale(glm::make_vec3(scale)) : glm::dmat4(1.0));
    localTransform *= (rotation ? glm::mat4_cast(glm::make_quat(rotation)) : glm::dmat4(1.0));
    glm::dmat4 transform = inTransform * localTransform;
    std::memcpy(out, glm::value_ptr(transform), 128);
}

// geom->setTransform(transform) (Viewer.cpp:258): the dmat4 becomes the mat4 parameter of
// TriangleArrayInstance::setTransform (VertexInstance.inl:54-58); meshUniformData.transform / transformInv as uploaded
void glm_pin_mesh_transform(const double* transform, float* out_transform, float* out_transformInv) {
    glm::mat4 t = glm::mat4(glm::dmat4(glm::make_mat4(transform)));
    glm::mat4 a = glm::transpose(t);
    glm::mat4 b = glm::inverse(t);
    std::memcpy(out_transform, glm::value_ptr(a), 64);
    std::memcpy(out_transformInv, glm::value_ptr(b), 64);
}
}

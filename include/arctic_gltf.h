/* arctic_gltf.h -- C ABI of the scene-loader stand-in.
 *
 * Replaces, on Linux and without assimp / stb, what App::load_scene does above the renderer boundary
 * (reference src/app.cpp:173-385): read a glTF 2.0 file and hand the renderer materials (three RGBA8 images each),
 * meshes (Vertex + uint32 index arrays) and objects (model matrix + mesh index).  The conventions of that function are
 * restated, quirks included:
 *   - one mesh per glTF primitive, materials in file order (assimp's glTF2 importer), triangles only;
 *   - aiProcess_FlipUVs: v -> 1 - v; tangents from the file's TANGENT attribute (bitangent = cross(n, t) * w) or, without
 *     it, computed per triangle from the UV gradients and orthogonalised against the normal (aiProcess_CalcTangentSpace,
 *     without assimp's cross-vertex smoothing);
 *   - missing textures fall back to a white image / a flat normal map (assets/white.png, assets/normal.png);
 *   - node matrices go through assimp_to_mat4 (app.cpp:540-564), which feeds assimp's row-major elements to glm's
 *     column-major constructor, i.e. TRANSPOSES them, and are accumulated as parent * child in that transposed form.
 * Supported: .gltf (JSON) with external or base64 buffers and .glb containers, float / normalised-integer attributes,
 * u8/u16/u32 indices, images by uri or bufferView: PNG (1-16 bit, grey / RGB / palette / alpha, non-interlaced) and
 * baseline JPEG (8 bit, 1 or 3 components, sampling up to 2x2, restart intervals; float IDCT and replicated chroma, so an
 * LSB or two away from stb_image's integer pipeline).  Not supported: progressive JPEG, sparse accessors, Draco.
 * Nothing here runs on the GPU; parity with assimp's output is unpinned (assimp is not available offline).
 */
#ifndef ARCTIC_GLTF_H
#define ARCTIC_GLTF_H
#include <stdint.h>
#include "arctic_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ArcticGltf ArcticGltf;

/* NULL on failure with a message in err. */
ArcticGltf *arctic_gltf_load(const char *path, char *err, uint64_t err_len);
void arctic_gltf_free(ArcticGltf *g);

uint64_t arctic_gltf_material_count(const ArcticGltf *g);
uint64_t arctic_gltf_mesh_count(const ArcticGltf *g);
uint64_t arctic_gltf_object_count(const ArcticGltf *g);

/* image k of material i: 0 diffuse, 1 normal, 2 metal-rough; RGBA8, row-major (what stbi_load(..., 4) returns) */
int arctic_gltf_material_image(const ArcticGltf *g, uint64_t i, int k, const uint8_t **rgba, uint32_t *w, uint32_t *h);
int arctic_gltf_mesh(const ArcticGltf *g, uint64_t i, const ArcticVertex **vertices, uint64_t *n_vertices,
                     const uint32_t **indices, uint64_t *n_indices, uint64_t *material);
const ArcticObject *arctic_gltf_objects(const ArcticGltf *g);

/* convenience: create_material / create_mesh for everything in the file, in order (what load_scene does). */
int arctic_gltf_upload(const ArcticGltf *g, ArcticRenderer *r);

/* the image decoders alone (tests): PNG or baseline JPEG by signature; returns a malloc'ed RGBA8 image, NULL on failure */
uint8_t *arctic_png_decode(const uint8_t *data, uint64_t size, uint32_t *w, uint32_t *h, char *err, uint64_t err_len);
void arctic_png_free(uint8_t *p);

#ifdef __cplusplus
}
#endif
#endif

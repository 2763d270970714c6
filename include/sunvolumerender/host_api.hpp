// host_api.hpp -- the reference's host-side class names over the C ABI of libsvr_hip.so.
//
// A C++ host written against SunVolumeRender (gui/canvas.cpp) includes this instead of
// pathtracer.h / raycasting.h / core/*.h and links -lsvr_hip instead of the CUDA objects.  The classes
// derive from the POD structs of svr_abi.h, so their layout is the reference's byte for byte
// (SURVEY.md 8(b)); only the HOST-side methods exist here -- the __device__ methods of the reference
// classes (GenerateRay, Intersect, operator(), ...) live in the HIP kernels.
//
// Vector types: if <glm/glm.hpp> was included first, glm::vec2/vec3/u8vec4 are used; otherwise minimal
// layout-compatible types are provided in namespace glm so host code compiles unchanged.
#ifndef SUNVOLUMERENDER_HOST_API_HPP
#define SUNVOLUMERENDER_HOST_API_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define SVR_ABI_NO_REFERENCE_PROTOTYPES
#include "../svr_abi.h"

#ifndef GLM_VERSION
namespace glm {
struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a, float b) : x(a), y(b) {} explicit vec2(float a) : x(a), y(a) {} };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
};
struct u8vec4 { uint8_t x, y, z, w; };
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(float s, const vec3& a) { return vec3(s / a.x, s / a.y, s / a.z); }
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 normalize(const vec3& a) { float s = 1.f / std::sqrt(dot(a, a)); return a * s; }
inline vec3 cross(const vec3& x, const vec3& y) { return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline float length(const vec3& a) { return std::sqrt(dot(a, a)); }
}  // namespace glm
#endif

static_assert(sizeof(glm::vec3) == 12 && sizeof(glm::vec2) == 8 && sizeof(glm::u8vec4) == 4, "vector layouts");

namespace svr_detail {
inline svr_vec3 to(const glm::vec3& v) { svr_vec3 r = {v.x, v.y, v.z}; return r; }
inline svr_vec2 to(const glm::vec2& v) { svr_vec2 r = {v.x, v.y}; return r; }
inline glm::vec3 from(const svr_vec3& v) { return glm::vec3(v.x, v.y, v.z); }
}  // namespace svr_detail

typedef uint64_t cudaTextureObject_t;   // opaque software-texture handle (svr_create_*_texture)

// core/geometry/cuda_bbox.h
class cudaBBox : public svr_bbox {
public:
    cudaBBox() : svr_bbox() {}
    cudaBBox(const glm::vec3& vmin_, const glm::vec3& vmax_) { Set(vmin_, vmax_); }
    void Set(const glm::vec3& vmin_, const glm::vec3& vmax_)
    {
        vmin = svr_detail::to(vmin_);
        vmax = svr_detail::to(vmax_);
        invSize = svr_detail::to(1.f / (vmax_ - vmin_));
    }
};

// core/cuda_volume.h (host-side methods)
class cudaVolume : public svr_volume {
public:
    cudaVolume() : svr_volume() { densityScale = 1.f; gradientFactor = 0.5f; x_clip = y_clip = z_clip = svr_vec2{-1.f, 1.f}; }
    void Set(const cudaBBox& box, const glm::vec3& sp, const cudaTextureObject_t& t)
    {
        bbox = box;
        spacing = svr_detail::to(sp);
        invSpacing = svr_detail::to(1.f / sp);
        tex = t;
    }
    void SetClipPlane(const glm::vec2& xc, const glm::vec2& yc, const glm::vec2& zc) { x_clip = svr_detail::to(xc); y_clip = svr_detail::to(yc); z_clip = svr_detail::to(zc); }
    void SetXClipPlane(const glm::vec2& c) { x_clip = svr_detail::to(c); }
    void SetYClipPlane(const glm::vec2& c) { y_clip = svr_detail::to(c); }
    void SetZClipPlane(const glm::vec2& c) { z_clip = svr_detail::to(c); }
    void SetDensityScale(float s = 1.f) { densityScale = s; }
    glm::vec3 GetSize() const { return 1.f / svr_detail::from(bbox.invSize); }
    void SetInvMaxMagnitude(float m) { invMaxMagnitude = m; }
    float GetInvMaxMagnitude() const { return invMaxMagnitude; }
    void SetGradientFactor(float g) { gradientFactor = g; }
    float GetGradientFactor() const { return gradientFactor; }
};

// core/cuda_transfer_function.h
class cudaTransferFunction : public svr_transfer_function {
public:
    cudaTransferFunction() : svr_transfer_function() {}
    void Set(const cudaTextureObject_t& t, float maxOpacity_) { tex = t; maxOpacity = maxOpacity_; }
    float GetMaxOpacity() const { return maxOpacity; }
};

// core/cuda_camera.h:35-63
class cudaCamera : public svr_camera {
public:
    cudaCamera() : svr_camera() {}
    cudaCamera(const glm::vec3& pos_, const glm::vec3& target, const glm::vec3& up, float fovx = 45.f, float apeture_ = 0.f,
               float focalLength_ = 0.f, float exposure_ = 1.f, unsigned int w_ = 640, unsigned int h_ = 480)
    {
        Setup(pos_, target, up, fovx, apeture_, focalLength_, exposure_, w_, h_);
    }
    void Setup(const glm::vec3& pos_, const glm::vec3& u_, const glm::vec3& v_, const glm::vec3& w_, float fovx, float apeture_,
               float focalLength_, float exposure_, unsigned int imageW_, unsigned int imageH_)
    {
        pos = svr_detail::to(pos_); u = svr_detail::to(u_); v = svr_detail::to(v_); w = svr_detail::to(w_);
        imageW = imageW_; imageH = imageH_;
        aspectRatio = (float)imageW / (float)imageH;
        tanFovxOverTwo = tanf(fovx * 0.5f * M_PI / 180.f);
        exposure = exposure_; focalLength = focalLength_; apeture = apeture_;
    }
    void Setup(const glm::vec3& pos_, const glm::vec3& target, const glm::vec3& up, float fovx, float apeture_, float focalLength_,
               float exposure_, unsigned int imageW_, unsigned int imageH_)
    {
        glm::vec3 w_ = glm::normalize(pos_ - target);
        glm::vec3 u_ = glm::cross(up, w_);
        glm::vec3 v_ = glm::cross(w_, u_);
        Setup(pos_, u_, v_, w_, fovx, apeture_, focalLength_, exposure_, imageW_, imageH_);
    }
};

// core/geometry/cuda_disk.h
class cudaDisk : public svr_disk {
public:
    cudaDisk() : svr_disk() {}
    cudaDisk(const glm::vec3& c, const glm::vec3& n, float r) { Set(c, n, r); }
    void Set(const glm::vec3& c, const glm::vec3& n, float r) { center = svr_detail::to(c); normal = svr_detail::to(n); radius = r; }
    float GetArea() const { return M_PI * radius * radius; }
};

// core/lights/cuda_arealight.h
class cudaAreaLight : public svr_area_light {
public:
    cudaAreaLight() : svr_area_light() {}
    void Set(const cudaDisk& d, const glm::vec3& c, float i) { disk = d; color = svr_detail::to(c); intensity = i; }
    void SetShape(const cudaDisk& d) { disk = d; }
    void SetColor(const glm::vec3& c) { color = svr_detail::to(c); }
    void SetIntensity(float i) { intensity = i; }
    void SetRadius(float r) { disk.radius = r; }
    void SetPosition(const glm::vec3& p) { disk.center = svr_detail::to(p); }
    void SetNormal(const glm::vec3& n) { disk.normal = svr_detail::to(n); }
    glm::vec3 GetColor() const { return svr_detail::from(color); }
    float GetIntensity() const { return intensity; }
    float GetRadius() const { return disk.radius; }
    glm::vec3 GetCenter() const { return svr_detail::from(disk.center); }
};

// core/lights/cuda_environment_light.h
class cudaEnvironmentLight : public svr_environment_light {
public:
    cudaEnvironmentLight() : svr_environment_light() {}
    void Set(cudaTextureObject_t t) { tex = t; intensity = 1.f; offset = svr_vec2{0.f, 0.f}; }
    void Set(const glm::vec3& radiance) { tex = 0; defaultRadiance = svr_detail::to(radiance); intensity = 1.f; offset = svr_vec2{0.f, 0.f}; }
    void SetIntensity(float i) { intensity = i; }
    void SetOffset(const glm::vec2& o) { offset = svr_detail::to(o); }
    cudaTextureObject_t Get() { return tex; }
};

// core/render_parameters.h
class RenderParams : public svr_render_params {
public:
    RenderParams() { traceDepth = 1; frameNo = 0; hdrBuffer = nullptr; }
    void SetupHDRBuffer(uint32_t w, uint32_t h) { svr_render_params_setup_hdr(this, w, h); }
    void Clear() { svr_render_params_clear(this); }
};

static_assert(sizeof(cudaBBox) == 36 && sizeof(cudaVolume) == 112 && sizeof(cudaTransferFunction) == 16 && sizeof(cudaCamera) == 76 &&
              sizeof(cudaDisk) == 28 && sizeof(cudaAreaLight) == 44 && sizeof(cudaEnvironmentLight) == 32 && sizeof(RenderParams) == 16,
              "reference POD layouts");

// the reference's device-layer entry points with the reference's own signatures (pathtracer.h:17-24, raycasting.h:8)
extern "C" void render_pathtracer(glm::u8vec4* img, const RenderParams& renderParams);
extern "C" void setup_volume(const cudaVolume& vol);
extern "C" void setup_transferfunction(const cudaTransferFunction& tf);
extern "C" void setup_camera(const cudaCamera& cam);
extern "C" void setup_env_lights(const cudaEnvironmentLight& light);
extern "C" void setup_area_lights(cudaAreaLight* lights, uint32_t n);
extern "C" void render_raycasting(glm::u8vec4* img, cudaVolume& volume, cudaTransferFunction& transferFunction, cudaCamera& camera, float stepSize);

#endif  // SUNVOLUMERENDER_HOST_API_HPP

// canvas.hpp -- the reference's host-side classes around the render entry points, headless, over the C ABI of
// libsvr_hip.so (include/svr_abi.h, include/svr_io.h).  Same class and method names, same call order:
//
//   VolumeReader       core/VolumeReader.{h,cpp}      Read(.mhd/.mha), CreateDeviceVolume(cudaVolume*), sizes
//   TransferFunction   gui/transferfunction.{h,cpp}   node lists -> 1024 x RGBA table + maxOpacity, .tf save / load
//   Lights             core/lights/lights.{h,cpp}     environment light (.hdr or constant), area-light list
//   Canvas             gui/canvas.{h,cpp}             owns the scene PODs; every setter re-uploads through setup_* and
//                                                     restarts the progressive render; paintGL() renders one frame
//
// What is gone is Qt/OpenGL: paintGL() renders into a device image owned by the Canvas (the reference maps a GL
// pixel buffer), mouse / keyboard handlers become Rotate / Translate / Zoom calls, and the 0-ms timer that drives
// progressive refinement is the caller's loop.  The image size is a constructor argument (the reference's WIDTH /
// HEIGHT macros, common.h:8-9).
#ifndef SUNVOLUMERENDER_CANVAS_HPP
#define SUNVOLUMERENDER_CANVAS_HPP

#include <string>
#include <vector>

#include "host_api.hpp"
#include "../svr_io.h"

// ---------------------------------------------------------------------------------------------------
// core/VolumeReader.h
// ---------------------------------------------------------------------------------------------------
class VolumeReader {
public:
    VolumeReader() = default;
    ~VolumeReader() { ClearDevice(); }
    VolumeReader(const VolumeReader&) = delete;                // owns a device texture
    VolumeReader& operator=(const VolumeReader&) = delete;

    // VolumeReader.cpp:13-77: the file is parsed on the host; cast, range, rescale, histogram and gradient-magnitude
    // maximum run on the GPU, and the texture is built at the same time (the reference does that in
    // CreateDeviceVolume; the result is the same object)
    void Read(std::string filename, int layout = SVR_LAYOUT_AUTO)
    {
        ClearDevice();
        histogram.assign(65536, 0u);
        svr_volume_info info;
        if (svr_load_mhd(filename.c_str(), layout, &loaded, &info, histogram.data(), (uint32_t)histogram.size()) != 0) {
            histogram.clear();
            return;                                   // svr_last_error() tells why (fatal mode has already exited)
        }
        histogram.resize(info.hist_bins < histogram.size() ? info.hist_bins : histogram.size());
        dim[0] = info.dim[0]; dim[1] = info.dim[1]; dim[2] = info.dim[2];
        spacing = glm::vec3(info.spacing[0], info.spacing[1], info.spacing[2]);
        maxMagnitude = info.maxMagnitude;
        range[0] = info.range[0]; range[1] = info.range[1];
        have = true;
    }

    // VolumeReader.cpp:174-185
    void CreateDeviceVolume(cudaVolume* volume)
    {
        if (!have) return;
        volume->bbox = loaded.bbox;
        volume->spacing = loaded.spacing;
        volume->invSpacing = loaded.invSpacing;
        volume->tex = loaded.tex;
        volume->SetInvMaxMagnitude(1.f / maxMagnitude);
    }

    glm::vec3 GetVolumeSize() { return glm::vec3(dim[0] * spacing.x, dim[1] * spacing.y, dim[2] * spacing.z); }
    float GetBoundingSphereRadius() { return glm::length(GetVolumeSize()) * 0.5f; }
    float GetElementBoundingSphereRadius() const { return glm::length(spacing) * 0.5f; }
    bool IsLoaded() const { return have; }

    std::vector<uint32_t> histogram;
    int dim[3] = {0, 0, 0};
    double range[2] = {0, 0};
    float maxMagnitude = 0.f;

private:
    void ClearDevice()
    {
        if (have && loaded.tex) svr_destroy_texture(loaded.tex);
        loaded = svr_volume();
        have = false;
    }
    glm::vec3 spacing;
    svr_volume loaded = svr_volume();
    bool have = false;
};

// ---------------------------------------------------------------------------------------------------
// gui/transferfunction.h (vtkPiecewiseFunction + vtkColorTransferFunction node lists included)
// ---------------------------------------------------------------------------------------------------
class TransferFunction {
public:
    static const int TABLE_SIZE = SVR_TF_TABLE_SIZE;

    TransferFunction() = default;
    ~TransferFunction() { if (compositeTex) svr_destroy_texture(compositeTex); }
    TransferFunction(const TransferFunction&) = delete;        // owns a device texture
    TransferFunction& operator=(const TransferFunction&) = delete;

    // vtkPiecewiseFunction::AddPoint / vtkColorTransferFunction::AddRGBPoint: sorted by x, same x replaces
    void AddPoint(double x, double y, double midpoint = 0.5, double sharpness = 0.0)
    {
        const double node[4] = {x, y, midpoint, sharpness};
        Insert(opacity, 4, node);
    }
    void AddRGBPoint(double x, double r, double g, double b, double midpoint = 0.5, double sharpness = 0.0)
    {
        const double node[6] = {x, r, g, b, midpoint, sharpness};
        Insert(color, 6, node);
    }
    void RemoveAllPoints() { opacity.clear(); color.clear(); }
    int GetOpacitySize() const { return (int)opacity.size() / 4; }
    int GetColorSize() const { return (int)color.size() / 6; }

    // constructor body + onOpacityTFChanged / onColorTFChanged (transferfunction.cpp:17-44, 128-175): rebuild the
    // composite table and maxOpacity, (re)upload the 1-D texture; returns the texture handle
    cudaTextureObject_t Update()
    {
        svr_tf_build_table(opacity.data(), GetOpacitySize(), color.data(), GetColorSize(), TABLE_SIZE, compositeTable, &maxOpacity);
        if (compositeTex) svr_update_tf_texture(compositeTex, compositeTable, TABLE_SIZE, 0);
        else compositeTex = svr_create_tf_texture(compositeTable, TABLE_SIZE, 0);
        return compositeTex;
    }
    cudaTextureObject_t GetCompositeTFTextureObject() const { return compositeTex; }
    float GetMaxOpacityValue() const { return maxOpacity; }

    // transferfunction.cpp:55-126, without the file dialogs
    bool SaveCurrentTFConfiguration(const std::string& filename) const
    {
        return svr_tf_save(filename.c_str(), opacity.data(), GetOpacitySize(), color.data(), GetColorSize()) == 0;
    }
    bool LoadExistingTFConfiguration(const std::string& filename)
    {
        int n = 0, m = 0;
        if (svr_tf_load(filename.c_str(), nullptr, &n, nullptr, &m) != 0) return false;      // counts only
        std::vector<double> o((size_t)n * 4), c((size_t)m * 6);
        if (svr_tf_load(filename.c_str(), o.data(), &n, c.data(), &m) != 0) return false;
        RemoveAllPoints();
        for (int i = 0; i < n; ++i) Insert(opacity, 4, &o[(size_t)i * 4]);
        for (int i = 0; i < m; ++i) Insert(color, 6, &c[(size_t)i * 6]);
        return true;
    }

    float compositeTable[SVR_TF_TABLE_SIZE * 4] = {0};

private:
    static void Insert(std::vector<double>& nodes, int stride, const double* node)
    {
        size_t i = 0, n = nodes.size() / stride;
        while (i < n && nodes[i * stride] < node[0]) ++i;
        if (i < n && nodes[i * stride] == node[0]) { for (int k = 0; k < stride; ++k) nodes[i * stride + k] = node[k]; return; }
        nodes.insert(nodes.begin() + (long)(i * stride), node, node + stride);
    }
    std::vector<double> opacity, color;
    cudaTextureObject_t compositeTex = 0;
    float maxOpacity = 0.f;
};

// ---------------------------------------------------------------------------------------------------
// core/lights/lights.h
// ---------------------------------------------------------------------------------------------------
class Lights {
public:
    Lights() { environmentLight.Set(glm::vec3(0.03f)); }                       // lights.cpp:11-14
    ~Lights() { if (envTex) svr_destroy_texture(envTex); }
    Lights(const Lights&) = delete;                            // owns a device texture
    Lights& operator=(const Lights&) = delete;

    void SetEnvironmentLight(std::string filename)                              // lights.cpp:31-75
    {
        cudaTextureObject_t old = envTex;
        svr_environment_light tmp = environmentLight;
        if (svr_load_env_map(filename.c_str(), &tmp) != 0) return;
        environmentLight.Set(tmp.tex);
        envTex = tmp.tex;
        if (old) svr_destroy_texture(old);
    }
    void SetEnvionmentLight(const glm::vec3& radiance) { environmentLight.Set(radiance); }    // sic, lights.cpp:77
    void SetEnvironmentLightIntensity(float intensity) { environmentLight.SetIntensity(intensity); }
    void SetEnvironmentLightOffset(const glm::vec2& offset) { environmentLight.SetOffset(offset); }
    void AddAreaLights(const cudaAreaLight& areaLight, const glm::vec3& tm)     // lights.cpp:92-103
    {
        if (areaLights.size() <= SVR_MAX_LIGHT_SOURCES) { areaLights.push_back(areaLight); transforms.push_back(tm); }
        else fprintf(stderr, "Exceed maximum number of light sources\n");
    }
    void RemoveLights(uint32_t idx)
    {
        if (!areaLights.empty()) { areaLights.erase(areaLights.begin() + idx); transforms.erase(transforms.begin() + idx); }
    }

    cudaEnvironmentLight environmentLight;
    std::vector<cudaAreaLight> areaLights;
    std::vector<glm::vec3> transforms;

private:
    cudaTextureObject_t envTex = 0;
};

// ---------------------------------------------------------------------------------------------------
// gui/canvas.h
// ---------------------------------------------------------------------------------------------------
enum RenderMode { RENDER_MODE_PATHTRACER, RENDER_MODE_RAYCASTING };

class Canvas {
public:
    Canvas(int width, int height) : WIDTH(width), HEIGHT(height)              // canvas.cpp:8-20
    {
        volumeReader = new VolumeReader();
        lights.SetEnvionmentLight(glm::vec3(1.f));
        lights.SetEnvironmentLightIntensity(0.5f);
        setup_env_lights(lights.environmentLight);
        renderParams.SetupHDRBuffer(WIDTH, HEIGHT);
        renderParams.traceDepth = 1;
        deviceVolume.SetGradientFactor(0.5f);
        img = (glm::u8vec4*)svr_device_malloc((size_t)WIDTH * HEIGHT * 4);
        view[0] = glm::vec3(1.f, 0.f, 0.f); view[1] = glm::vec3(0.f, 1.f, 0.f); view[2] = glm::vec3(0.f, 0.f, 1.f);
    }
    ~Canvas()
    {
        svr_device_synchronize();
        renderParams.Clear();
        if (img) svr_device_free(img);
        delete volumeReader;
    }
    Canvas(const Canvas&) = delete;
    Canvas& operator=(const Canvas&) = delete;

    void LoadVolume(std::string filename)                                       // canvas.cpp:27-41
    {
        volumeReader->Read(filename);
        if (!volumeReader->IsLoaded()) return;
        volumeReader->CreateDeviceVolume(&deviceVolume);
        deviceVolume.SetClipPlane(glm::vec2(-1.f, 1.f), glm::vec2(-1.f, 1.f), glm::vec2(-1.f, 1.f));
        deviceVolume.SetDensityScale(1.f);
        setup_volume(deviceVolume);
        ZoomToExtent();
        view[0] = glm::vec3(1.f, 0.f, 0.f); view[1] = glm::vec3(0.f, 1.f, 0.f); view[2] = glm::vec3(0.f, 0.f, 1.f);
        cameraTranslate = glm::vec2(0.f, 0.f);
        UpdateCamera();
        ready = true;
        ReStartRender();
    }

    void ReStartRender() { renderParams.frameNo = 0; }                          // canvas.h:43-47

    void SetTransferFunction(const cudaTextureObject_t& tex, float maxOpacity)  // canvas.h:49-54
    {
        transferFunction.Set(tex, maxOpacity);
        setup_transferfunction(transferFunction);
        ReStartRender();
    }
    void SetDensityScale(double s) { deviceVolume.SetDensityScale((float)s); setup_volume(deviceVolume); ReStartRender(); }
    void SetGradientFactor(double g) { deviceVolume.SetGradientFactor((float)g); setup_volume(deviceVolume); ReStartRender(); }
    void SetScatterTimes(double val) { renderParams.traceDepth = (uint32_t)val; ReStartRender(); }
    void SetRenderMode(RenderMode mode) { renderMode = mode; ReStartRender(); }

    // lights, canvas.h:96-133
    void SetEnvLightBackground(const glm::vec3& color) { lights.SetEnvionmentLight(color); setup_env_lights(lights.environmentLight); ReStartRender(); }
    void SetEnvLightMap(std::string filename) { lights.SetEnvironmentLight(filename); setup_env_lights(lights.environmentLight); ReStartRender(); }
    void SetEnvLightOffset(const glm::vec2& offset) { lights.SetEnvironmentLightOffset(offset); setup_env_lights(lights.environmentLight); ReStartRender(); }
    void SetEnvLightIntensity(float intensity) { lights.SetEnvironmentLightIntensity(intensity); setup_env_lights(lights.environmentLight); ReStartRender(); }
    void SetAreaLights() { setup_area_lights(lights.areaLights.data(), (uint32_t)lights.areaLights.size()); ReStartRender(); }

    // camera, canvas.h:136-162
    void SetFOV(float f) { fov = f; UpdateCamera(); ReStartRender(); }
    void SetApeture(float a) { apeture = a; UpdateCamera(); ReStartRender(); }
    void SetFocalLength(float f) { focalLength = f; UpdateCamera(); ReStartRender(); }
    void SetExposure(float e) { exposure = e; UpdateCamera(); ReStartRender(); }

    // clip planes, canvas.h:165-184
    void SetXClipPlane(double mn, double mx) { deviceVolume.SetXClipPlane(glm::vec2(float(mn), float(mx))); setup_volume(deviceVolume); ReStartRender(); }
    void SetYClipPlane(double mn, double mx) { deviceVolume.SetYClipPlane(glm::vec2(float(mn), float(mx))); setup_volume(deviceVolume); ReStartRender(); }
    void SetZClipPlane(double mn, double mx) { deviceVolume.SetZClipPlane(glm::vec2(float(mn), float(mx))); setup_volume(deviceVolume); ReStartRender(); }

    // what the mouse / wheel / arrow-key handlers do to the view (canvas.cpp:119-226): rotate the view basis about
    // an axis given in view space, pan, dolly
    void Rotate(float angleDegrees, const glm::vec3& axis)
    {
        const float a = angleDegrees * 3.14159265358979323846f / 180.f, c = std::cos(a), s = std::sin(a);
        const glm::vec3 k = glm::normalize(axis);
        const glm::vec3 world = view[0] * k.x + view[1] * k.y + view[2] * k.z;
        for (int i = 0; i < 3; ++i)                                             // Rodrigues
            view[i] = view[i] * c + glm::cross(world, view[i]) * s + world * (glm::dot(world, view[i]) * (1.f - c));
        UpdateCamera();
        ReStartRender();
    }
    void Translate(const glm::vec2& delta) { cameraTranslate = glm::vec2(cameraTranslate.x + delta.x, cameraTranslate.y + delta.y); UpdateCamera(); ReStartRender(); }
    void Zoom(float delta) { eyeDist += delta; UpdateCamera(); ReStartRender(); }

    // paintGL's render branch, canvas.cpp:70-116
    void paintGL()
    {
        if (!ready) return;
        if (renderMode == RENDER_MODE_RAYCASTING)
            render_raycasting(img, deviceVolume, transferFunction, camera, volumeReader->GetElementBoundingSphereRadius());
        else {
            render_pathtracer(img, renderParams);
            if (renderParams.frameNo == 0 && dumpFirstFrame) SaveImage("0.tga");   // canvas.cpp:97-104
        }
        svr_device_synchronize();
        renderParams.frameNo++;
    }

    // extension: n progressive frames in one launch group
    void paintFrames(uint32_t n)
    {
        if (!ready || renderMode != RENDER_MODE_PATHTRACER) return;
        svr_render_pathtracer_frames(img, &renderParams, n);
        renderParams.frameNo += n;
    }

    bool SaveImage(const std::string& filename)
    {
        std::vector<uint8_t> host((size_t)WIDTH * HEIGHT * 4);
        if (svr_memcpy_d2h(host.data(), img, host.size()) != 0) return false;
        return svr_tga_write(filename.c_str(), WIDTH, HEIGHT, host.data()) == 0;
    }
    void ReadImage(std::vector<uint8_t>& host)
    {
        host.resize((size_t)WIDTH * HEIGHT * 4);
        svr_memcpy_d2h(host.data(), img, host.size());
    }

    uint32_t FrameNo() const { return renderParams.frameNo; }
    void* HdrBuffer() const { return renderParams.hdrBuffer; }               // the accumulator (device): what svr_assemble_frame sends
    const cudaCamera& Camera() const { return camera; }
    const cudaVolume& Volume() const { return deviceVolume; }

    VolumeReader* volumeReader;
    Lights lights;
    bool dumpFirstFrame = false;                   // the reference always writes 0.tga; off by default here

    const int WIDTH, HEIGHT;

private:
    void ZoomToExtent()                                                         // canvas.cpp:191-197
    {
        glm::vec3 extent = volumeReader->GetVolumeSize();
        float maxSpan = fmaxf(extent.x, fmaxf(extent.y, extent.z));
        maxSpan *= 1.5f;
        eyeDist = maxSpan / (2 * tan((fov * 0.5f) * 0.01745329251994329576923690768489f));
    }
    void UpdateCamera()                                                         // canvas.cpp:178-188
    {
        const glm::vec3 u = view[0], v = view[1], w = view[2];
        const glm::vec3 pos = w * eyeDist - u * cameraTranslate.x - v * cameraTranslate.y;
        camera.Setup(pos, u, v, w, fov, apeture, focalLength, exposure, WIDTH, HEIGHT);
        setup_camera(camera);
    }

    bool ready = false;
    glm::u8vec4* img = nullptr;
    float exposure = 1.f, apeture = 0.f, fov = 45.f, focalLength = 1.f, eyeDist = 0.f;
    glm::vec2 cameraTranslate;
    glm::vec3 view[3];                             // rows of the reference's viewMat: camera u, v, w
    RenderParams renderParams;
    cudaCamera camera;
    cudaVolume deviceVolume;
    cudaTransferFunction transferFunction;
    RenderMode renderMode = RENDER_MODE_RAYCASTING;
};

#endif  // SUNVOLUMERENDER_CANVAS_HPP

// headless_canvas.cpp -- C++ host replaying the reference's Canvas protocol (gui/canvas.cpp:8-41, 63-117)
// against libsvr_hip.so: synthetic sphere volume, GUI-default transfer function and light, N progressive
// frames, writes frame.ppm.  Build: see examples/Makefile.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sunvolumerender/host_api.hpp"

int main(int argc, char** argv)
{
    const int N = 64, W = 256, H = 256;
    const int frames = argc > 1 ? atoi(argv[1]) : 16;
    if (svr_init(0)) return 1;

    // VolumeReader::Read + CreateDeviceVolume (core/VolumeReader.cpp:138-185), synthetic data
    std::vector<uint16_t> vox((size_t)N * N * N);
    for (int z = 0; z < N; ++z)
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x) {
                float fx = (x + 0.5f) / N - 0.5f, fy = (y + 0.5f) / N - 0.5f, fz = (z + 0.5f) / N - 0.5f;
                float r = std::sqrt(fx * fx + fy * fy + fz * fz);
                float d = r < 0.35f ? 1.f - r / 0.7f : 0.f;
                vox[((size_t)z * N + y) * N + x] = (uint16_t)(d * 65535.f + 0.5f);
            }
    cudaTextureObject_t volTex = svr_create_volume_texture(vox.data(), N, N, N, 0, SVR_LAYOUT_AUTO);
    glm::vec3 extent(N * 1.f, N * 1.f, N * 1.f);
    glm::vec3 vmax = extent - extent * 0.5f;
    cudaVolume deviceVolume;
    deviceVolume.Set(cudaBBox(-vmax, vmax), glm::vec3(1.f), volTex);
    deviceVolume.SetInvMaxMagnitude(1.f / 3000.f);
    deviceVolume.SetGradientFactor(0.5f);                                    // canvas.cpp:19
    deviceVolume.SetClipPlane(glm::vec2(-1.f, 1.f), glm::vec2(-1.f, 1.f), glm::vec2(-1.f, 1.f));
    deviceVolume.SetDensityScale(1.f);

    // TransferFunction ctor (gui/transferfunction.cpp:17-44) with the GUI default opacity ramp (mainwindow.cpp:51-55)
    std::vector<float> table(SVR_TF_TABLE_SIZE * 4);
    for (int i = 0; i < SVR_TF_TABLE_SIZE; ++i) {
        float x = i / (SVR_TF_TABLE_SIZE - 1.f);
        table[4 * i + 0] = 0.8f; table[4 * i + 1] = 0.5f + 0.4f * x; table[4 * i + 2] = 0.3f;
        table[4 * i + 3] = x < 0.1f ? 5.f * x : 0.5f;
    }
    cudaTransferFunction transferFunction;
    transferFunction.Set(svr_create_tf_texture(table.data(), SVR_TF_TABLE_SIZE, 0), 0.5f);

    // Canvas::Canvas (canvas.cpp:8-20)
    cudaEnvironmentLight env;
    env.Set(glm::vec3(1.f));
    env.SetIntensity(0.5f);
    setup_env_lights(env);
    RenderParams renderParams;
    renderParams.SetupHDRBuffer(W, H);
    renderParams.traceDepth = 1;

    setup_transferfunction(transferFunction);                                 // mainwindow.cpp:27
    setup_volume(deviceVolume);                                               // canvas.cpp:33
    float eyeDist = 1.5f * N / (2.f * std::tan(22.5f * 3.14159265f / 180.f)); // ZoomToExtent, canvas.cpp:191-197
    cudaCamera camera;
    camera.Setup(glm::vec3(0.f, 0.f, eyeDist), glm::vec3(0.f), glm::vec3(0.f, 1.f, 0.f), 45.f, 0.f, 1.f, 1.f, W, H);
    setup_camera(camera);
    cudaAreaLight light;                                                      // mainwindow.cpp:229-238
    float dist = glm::length(extent) * 0.5f * 1.5f + 1.f;
    light.Set(cudaDisk(glm::vec3(0.f, dist, 0.f), glm::vec3(0.f, -1.f, 0.f), 10.f), glm::vec3(1.f), 500.f);
    setup_area_lights(&light, 1);

    glm::u8vec4* img = (glm::u8vec4*)svr_device_malloc((size_t)W * H * 4);
    for (int f = 0; f < frames; ++f) {                                        // paintGL, canvas.cpp:96-116
        render_pathtracer(img, renderParams);
        svr_device_synchronize();
        renderParams.frameNo++;
    }
    std::vector<uint8_t> host((size_t)W * H * 4);
    svr_memcpy_d2h(host.data(), img, host.size());
    FILE* fp = fopen("frame.ppm", "wb");
    fprintf(fp, "P6\n%d %d\n255\n", W, H);
    for (int y = H - 1; y >= 0; --y)
        for (int x = 0; x < W; ++x) fwrite(&host[4 * ((size_t)y * W + x)], 1, 3, fp);
    fclose(fp);
    printf("rendered %d frames on %s -> frame.ppm\n", frames, svr_device_info());
    svr_device_free(img);
    renderParams.Clear();
    svr_shutdown();
    return 0;
}

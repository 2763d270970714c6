// render_mhd.cpp -- the reference application's start-up sequence (main.cpp / gui/mainwindow.cpp:22-62, 229-238)
// without the GUI: load a MetaImage volume, the GUI-default (or a saved .tf) transfer function, one area light, an
// optional .hdr environment map; render N progressive frames; write the image as TGA.
//
//   render_mhd <volume.mhd> [-tf file.tf] [-env map.hdr] [-frames N] [-depth D] [-size W H] [-raycast] [-o out.tga]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "sunvolumerender/canvas.hpp"

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s volume.mhd [-tf f.tf] [-env m.hdr] [-frames N] [-depth D] [-size W H] [-raycast] [-o out.tga]\n", argv[0]); return 2; }
    std::string volume = argv[1], tfFile, envFile, out = "frame.tga";
    int frames = 16, depth = 1, W = 640, H = 640;                  // common.h:8-9
    bool raycast = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "-tf") && i + 1 < argc) tfFile = argv[++i];
        else if (!strcmp(argv[i], "-env") && i + 1 < argc) envFile = argv[++i];
        else if (!strcmp(argv[i], "-frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-depth") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-size") && i + 2 < argc) { W = atoi(argv[i + 1]); H = atoi(argv[i + 2]); i += 2; }
        else if (!strcmp(argv[i], "-raycast")) raycast = true;
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (svr_init(0)) return 1;
    {
        Canvas canvas(W, H);

        // MainWindow::MainWindow, mainwindow.cpp:22-62: default opacity ramp and colour map, or a saved configuration
        TransferFunction tf;
        if (!tfFile.empty()) {
            if (!tf.LoadExistingTFConfiguration(tfFile)) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        } else {
            tf.AddPoint(0.0, 0.0);
            for (int i = 1; i <= 10; ++i) tf.AddPoint(0.1 * i, 0.5);
            tf.AddRGBPoint(0.0, 69 / 255.0, 199 / 255.0, 186 / 255.0);
            tf.AddRGBPoint(0.2, 172 / 255.0, 3 / 255.0, 57 / 255.0);
            tf.AddRGBPoint(0.4, 169 / 255.0, 83 / 255.0, 58 / 255.0);
            tf.AddRGBPoint(0.6, 43 / 255.0, 32 / 255.0, 161 / 255.0);
            tf.AddRGBPoint(0.8, 247 / 255.0, 158 / 255.0, 97 / 255.0);
            tf.AddRGBPoint(1.0, 183 / 255.0, 7 / 255.0, 140 / 255.0);
        }
        const cudaTextureObject_t tfTex = tf.Update();                // (argument evaluation order is unspecified)
        canvas.SetTransferFunction(tfTex, tf.GetMaxOpacityValue());

        canvas.LoadVolume(volume);
        if (!canvas.volumeReader->IsLoaded()) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        printf("%s: %d x %d x %d, range %.0f..%.0f, max gradient magnitude %.0f, %zu histogram bins\n", volume.c_str(),
               canvas.volumeReader->dim[0], canvas.volumeReader->dim[1], canvas.volumeReader->dim[2], canvas.volumeReader->range[0],
               canvas.volumeReader->range[1], canvas.volumeReader->maxMagnitude, canvas.volumeReader->histogram.size());

        // the "add light" dialog's defaults (mainwindow.cpp:229-238): a disk above the volume, facing down
        cudaAreaLight light;
        float dist = canvas.volumeReader->GetBoundingSphereRadius() * 1.5f + 1.f;
        light.Set(cudaDisk(glm::vec3(0.f, dist, 0.f), glm::vec3(0.f, -1.f, 0.f), 10.f), glm::vec3(1.f), 500.f);
        canvas.lights.AddAreaLights(light, glm::vec3(0.f, 0.f, dist));
        canvas.SetAreaLights();
        if (!envFile.empty()) {
            canvas.SetEnvLightMap(envFile);
            svr_set_option(SVR_OPT_ENV_ON_ESCAPE, 1);
        }
        canvas.SetScatterTimes(depth);
        canvas.SetRenderMode(raycast ? RENDER_MODE_RAYCASTING : RENDER_MODE_PATHTRACER);

        for (int f = 0; f < (raycast ? 1 : frames); ++f) canvas.paintGL();
        if (!canvas.SaveImage(out)) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        printf("%u frame(s) on %s -> %s\n", canvas.FrameNo(), svr_device_info(), out.c_str());
    }
    svr_shutdown();
    return 0;
}

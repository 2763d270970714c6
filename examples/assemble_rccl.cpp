// assemble_rccl.cpp -- the multi-GPU protocol of DESIGN.md section 7 for a C++ host, without Python: one process per GPU,
// interleaved row strips, one RCCL exchange per output frame (svr_assemble_frame), tone map of the assembled frame on rank 0.
//
//   hipcc -I../include assemble_rccl.cpp -L../sunvolumerender_amd/lib -lsvr_hip -lrccl -o assemble_rccl
//   launch N copies with RANK / WORLD_SIZE / LOCAL_RANK in the environment (mpirun, torchrun --no-python, a shell loop); the
//   second argument is a file path all ranks see, for the ncclUniqueId (rank 0 writes it, the others wait for it).
//
// The repository only COMPILES this file (tests/test_abi.py): the boxes it is developed on have one GPU, and RCCL refuses two
// ranks on one device.  The device half of the exchange (pack / unpack) is tested in tests/test_dist_gpu.py, the index maths
// in tests/test_dist_cpu.py, and the same protocol end to end over torch.distributed in bench.py --gpus N.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "sunvolumerender/canvas.hpp"

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

int main(int argc, char** argv)
{
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    if (argc < 2) { fprintf(stderr, "usage: assemble_rccl volume.mhd [id-file]\n"); return 2; }
    const std::string id_path = argc > 2 ? argv[2] : "/tmp/svr_nccl_id";
    if (svr_init(local)) return 1;

    // communicator: rank 0 creates the id, the others wait for the file
    ncclComm_t comm = nullptr;
    if (world > 1) {
        ncclUniqueId id;
        if (rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
            FILE* f = fopen((id_path + ".tmp").c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 1;
            fclose(f);
            rename((id_path + ".tmp").c_str(), id_path.c_str());
        } else {
            FILE* f = nullptr;
            while (!(f = fopen(id_path.c_str(), "rb"))) {}
            if (fread(&id, sizeof id, 1, f) != 1) return 1;
            fclose(f);
        }
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
    }
    {
        // the reference's start-up sequence (gui/mainwindow.cpp:22-62, gui/canvas.cpp:27-41) on every rank: the full scene, its own strips
        const uint32_t W = 1024, H = 1024, strip = 16;
        Canvas canvas((int)W, (int)H);
        TransferFunction tf;
        tf.AddPoint(0.0, 0.0);
        for (int i = 1; i <= 10; ++i) tf.AddPoint(0.1 * i, 0.5);
        tf.AddRGBPoint(0.0, 69 / 255.0, 199 / 255.0, 186 / 255.0);
        tf.AddRGBPoint(1.0, 183 / 255.0, 7 / 255.0, 140 / 255.0);
        const cudaTextureObject_t tfTex = tf.Update();
        canvas.SetTransferFunction(tfTex, tf.GetMaxOpacityValue());
        canvas.LoadVolume(argv[1]);
        if (!canvas.volumeReader->IsLoaded()) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        cudaAreaLight light;
        const float dist = canvas.volumeReader->GetBoundingSphereRadius() * 1.5f + 1.f;
        light.Set(cudaDisk(glm::vec3(0.f, dist, 0.f), glm::vec3(0.f, -1.f, 0.f), 10.f), glm::vec3(1.f), 500.f);
        canvas.lights.AddAreaLights(light, glm::vec3(0.f, 0.f, dist));
        canvas.SetAreaLights();
        canvas.SetRenderMode(RENDER_MODE_PATHTRACER);
        svr_set_row_shard(strip, (uint32_t)rank, (uint32_t)world);
        svr_set_option(SVR_OPT_SKIP_TONEMAP, world > 1);          // a rank's own strips are never shown

        void* frame = rank == 0 ? svr_device_malloc((size_t)W * H * 12) : nullptr;
        void* img = rank == 0 ? svr_device_malloc((size_t)W * H * 4) : nullptr;
        for (int step = 0; step < 4; ++step) {
            canvas.paintFrames(64);                                // 64 more samples per pixel on this rank's rows
            // one collective per output: strips -> rank 0, then the tone map of the whole frame there
            if (svr_assemble_frame(comm, frame, canvas.HdrBuffer(), W, H, strip, (uint32_t)rank, (uint32_t)world, 0)) return 1;
            if (rank == 0 && svr_hdr_to_ldr_frame(img, frame, W, H)) return 1;
        }
        svr_device_synchronize();
        if (rank == 0) {
            std::vector<uint8_t> host((size_t)W * H * 4);
            svr_memcpy_d2h(host.data(), img, host.size());
            svr_tga_write("assembled.tga", (int)W, (int)H, host.data());
            printf("rank 0: assembled.tga written (%d ranks)\n", world);
            svr_device_free(frame);
            svr_device_free(img);
        }
    }
    if (comm) ncclCommDestroy(comm);
    svr_shutdown();
    return 0;
}

// ref_driver.cpp -- drives the REFERENCE's own ITMLib CPU engine (compiled from
// /root/reference by oracle/ref_build.sh) over a synthetic RGB-D sequence and dumps
// everything the TSDF parity tests compare against.  TEST INFRASTRUCTURE ONLY.
//
// This file is original code; it only *uses* the reference's public C++ API the way
// slam/InfiniTAM_tools.cpp:3-67 and slam/slam_pipeline.cpp:362-371 do
// (ITMBasicEngine ctor, turnOffTracking, gtC2wPoses, ProcessFrame, runRaycast,
// GetFreeImage/GetFreeVertex).  Private render-state members are read through the
// usual `#define private public` test hack -- no reference source is modified.
//
// usage: itm_ref <input.bin> <output.bin>          dump mode (parity fixtures)
//        itm_ref <input.bin> <output.bin> track    same dump with the default tracker ON (poses estimated, not given)
//        itm_ref <input.bin> - time              timing mode: ProcessFrame loop only, prints one JSON line
//        itm_ref <input.bin> - timetrack         timing mode with the default tracker ON
//                                                (frames after the first; the CPU baseline of bench.py)
//        itm_ref <input.bin> <output.bin> mesh   dump + the CPU meshing engine's triangles of the final scene ("mesh"
//                                                chunk: 9 floats per triangle) and, if GPS_REF_SAVE_DIR is set, the
//                                                scene's SaveToDirectory files (voxel.dat, alloc.dat, vba.txt, hash.dat,
//                                                excess.dat, last.txt) and the mesh's WritePLY output (mesh.ply) there
//        itm_ref mcprobe <output.bin>            marching-cubes case probe: for each of the 256 sign configurations of one
//                                                cube, the edges (0..11) of the triangles MeshScene emits, in order
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include <algorithm>
#include <stdexcept>
#include <limits>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define private public
#define protected public
#include "ITMLib/ITMLibDefines.h"
#include "ITMLib/Core/ITMBasicEngine.h"
#include "ITMLib/Objects/RenderStates/ITMRenderState_VH.h"
#include "ITMLib/Engines/Meshing/CPU/ITMMeshingEngine_CPU.h"
#include "ITMLib/Engines/Meshing/CPU/ITMMeshingEngine_CPU.tpp"
#undef private
#undef protected

using namespace ITMLib;

static FILE *g_out;

static void chunk(const char *name, int frame, const void *data, int64_t nbytes) {
    if (!g_out) return;  // timing mode
    char nm[32];
    memset(nm, 0, sizeof(nm));
    strncpy(nm, name, 31);
    fwrite(nm, 1, 32, g_out);
    int32_t f = frame;
    fwrite(&f, 4, 1, g_out);
    fwrite(&nbytes, 8, 1, g_out);
    if (nbytes) fwrite(data, 1, (size_t)nbytes, g_out);
}

static uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

struct Header {
    int32_t magic, W, H, nframes, nfree, dump_vba_every;
    float fx, fy, cx, cy, voxel, mu, vfmin, vfmax;
};

typedef ITMBasicEngine<ITMVoxel, ITMVoxelIndex> Engine;

static void dump_scene(Engine *eng, int frame, bool full_vba) {
    ITMScene<ITMVoxel, ITMVoxelIndex> *scene = eng->GetScene();
    const ITMHashEntry *ht = scene->index.GetEntries();
    const ITMVoxel *vba = scene->localVBA.GetVoxelBlocks();
    const int total = ITMVoxelBlockHash::noTotalEntries;
    std::vector<int32_t> rows;
    uint32_t crc = 0;
    std::vector<unsigned char> blocks;
    for (int i = 0; i < total; i++) {
        const ITMHashEntry &e = ht[i];
        if (e.ptr == -2 && e.offset == 0 && e.pos.x == 0 && e.pos.y == 0 && e.pos.z == 0) continue;
        rows.push_back(i); rows.push_back(e.pos.x); rows.push_back(e.pos.y); rows.push_back(e.pos.z);
        rows.push_back(e.offset); rows.push_back(e.ptr);
        if (e.ptr >= 0) {
            const unsigned char *b = (const unsigned char *)(vba + (size_t)e.ptr * SDF_BLOCK_SIZE3);
            for (int v = 0; v < SDF_BLOCK_SIZE3; v++) crc = crc32_update(crc, b + v * sizeof(ITMVoxel), 7);  // skip the pad byte
            if (full_vba) blocks.insert(blocks.end(), b, b + sizeof(ITMVoxel) * SDF_BLOCK_SIZE3);
        }
    }
    chunk("hash", frame, rows.data(), (int64_t)rows.size() * 4);
    chunk("vba_crc", frame, &crc, 4);
    if (full_vba) chunk("vba", frame, blocks.data(), (int64_t)blocks.size());
}

typedef ITMMeshingEngine_CPU<ITMVoxel, ITMVoxelIndex> MeshEngine;

static void dump_mesh(ITMScene<ITMVoxel, ITMVoxelIndex> *scene, int frame, const char *save_dir) {
    ITMMesh mesh(MEMORYDEVICE_CPU, 1u << 22);
    MeshEngine me;
    me.MeshScene(&mesh, scene);
    const ITMMesh::Triangle *t = mesh.triangles->GetData(MEMORYDEVICE_CPU);
    std::vector<float> pos((size_t)mesh.noTotalTriangles * 9);
    for (uint i = 0; i < mesh.noTotalTriangles; i++) {
        const Vector3f *p[3] = {&t[i].p0, &t[i].p1, &t[i].p2};
        for (int k = 0; k < 3; k++) { pos[i * 9 + k * 3] = p[k]->x; pos[i * 9 + k * 3 + 1] = p[k]->y; pos[i * 9 + k * 3 + 2] = p[k]->z; }
    }
    chunk("mesh", frame, pos.data(), (int64_t)pos.size() * 4);
    if (save_dir) {
        std::string d(save_dir);
        if (d.back() != '/') d += '/';
        scene->SaveToDirectory(d);
        mesh.WritePLY((d + "mesh.ply").c_str());
    }
}

// One cube (local voxel 3,3,3 of block 0,0,0) with the sign pattern of every case in turn; all other voxels of the block are
// positive, so the neighbouring cubes emit triangles too -- only triangles whose three vertices lie on edges of OUR cube are
// kept (a neighbour's triangle always has a vertex off our cube for non-zero sdf values).
static int mc_probe(const char *out_path) {
    g_out = fopen(out_path, "wb");
    ITMSceneParams params(0.02f, 100, 1.0f, 0.2f, 3.0f, false);  // voxel size 1: vertices come out in voxel units
    ITMScene<ITMVoxel, ITMVoxelIndex> *scene = new ITMScene<ITMVoxel, ITMVoxelIndex>(&params, false, MEMORYDEVICE_CPU);
    ITMHashEntry *ht = scene->index.GetEntries();
    ITMHashEntry empty;
    memset(&empty, 0, sizeof(empty));
    empty.ptr = -2;
    for (int i = 0; i < scene->index.noTotalEntries; i++) ht[i] = empty;
    ht[0].ptr = 0;  // block (0,0,0) hashes to bucket 0
    ITMVoxel *vb = scene->localVBA.GetVoxelBlocks();
    static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int edge_ends[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    std::vector<int8_t> table(256 * 16, -1);
    ITMMesh mesh(MEMORYDEVICE_CPU, 1u << 16);
    MeshEngine me;
    const int B = 3;
    for (int c = 0; c < 256; c++) {
        for (int v = 0; v < SDF_BLOCK_SIZE3; v++) { vb[v] = ITMVoxel(); vb[v].sdf = 16000; vb[v].w_depth = 1; }
        for (int k = 0; k < 8; k++)
            if (c & (1 << k)) {
                const int x = B + corner[k][0], y = B + corner[k][1], z = B + corner[k][2];
                vb[x + y * SDF_BLOCK_SIZE + z * SDF_BLOCK_SIZE * SDF_BLOCK_SIZE].sdf = -8000;
            }
        me.MeshScene(&mesh, scene);
        const ITMMesh::Triangle *t = mesh.triangles->GetData(MEMORYDEVICE_CPU);
        int n = 0;
        for (uint i = 0; i < mesh.noTotalTriangles; i++) {
            const Vector3f *p[3] = {&t[i].p0, &t[i].p1, &t[i].p2};
            int e[3];
            bool ours = true;
            for (int k = 0; k < 3 && ours; k++) {
                e[k] = -1;
                for (int ed = 0; ed < 12; ed++) {
                    const int *a = corner[edge_ends[ed][0]], *b = corner[edge_ends[ed][1]];
                    // on edge ed: the two coordinates the end points share are equal to them, the third lies strictly between
                    bool on = true;
                    const float q[3] = {p[k]->x - B, p[k]->y - B, p[k]->z - B};
                    for (int d = 0; d < 3; d++) {
                        if (a[d] == b[d]) on = on && q[d] == (float)a[d];
                        else on = on && q[d] > 0.f && q[d] < 1.f;
                    }
                    if (on) { e[k] = ed; break; }
                }
                if (e[k] < 0) ours = false;
            }
            if (!ours) continue;
            if (n + 3 > 15) { fprintf(stderr, "case %d: more than 5 triangles\n", c); return 4; }
            for (int k = 0; k < 3; k++) table[c * 16 + n++] = (int8_t)e[k];
        }
    }
    chunk("mc_table", -1, table.data(), (int64_t)table.size());
    fclose(g_out);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    if (std::string(argv[1]) == "mcprobe") return mc_probe(argv[2]);
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror("input"); return 2; }
    g_out = std::string(argv[2]) == "-" ? nullptr : fopen(argv[2], "wb");
    Header h;
    if (fread(&h, sizeof(h), 1, in) != 1 || h.magic != 0x47505331) { fprintf(stderr, "bad header\n"); return 2; }
    const int P = h.W * h.H;
    int32_t sizes[2] = {(int32_t)sizeof(ITMVoxel), (int32_t)sizeof(ITMHashEntry)};
    chunk("sizeof", -1, sizes, 8);

    ITMRGBDCalib calib;
    calib.intrinsics_rgb.SetFrom(h.W, h.H, h.fx, h.fy, h.cx, h.cy);
    calib.intrinsics_d = calib.intrinsics_rgb;
    calib.disparityCalib.SetStandard();

    ITMLibSettings *settings = new ITMLibSettings();
    settings->deviceType = ITMLibSettings::DEVICE_CPU;
    settings->createMeshingEngine = false;
    settings->sceneParams.voxelSize = h.voxel;
    settings->sceneParams.mu = h.mu;
    settings->sceneParams.viewFrustum_min = h.vfmin;
    settings->sceneParams.viewFrustum_max = h.vfmax;

    Vector2i dims(h.W, h.H);
    Engine *eng = new Engine(settings, calib, dims, dims);
    // "track" mode keeps the default tracker of ITMLibSettings (depth-only extended tracker) active; otherwise poses come
    // from gtC2wPoses exactly as slam/InfiniTAM_tools.cpp:59-62 sets the engine up for use_gt_pose: true
    const std::string mode = argc >= 4 ? std::string(argv[3]) : std::string();
    const bool track_mode = mode == "track" || mode == "timetrack";
    if (!track_mode) eng->turnOffTracking();

    std::vector<ITMUChar4Image *> rgbs(h.nframes);
    std::vector<ITMShortImage *> depths(h.nframes);
    std::vector<ORUtils::Matrix4<float> *> poses(h.nframes);
    for (int f = 0; f < h.nframes; f++) {
        rgbs[f] = new ITMUChar4Image(dims, true, false);
        depths[f] = new ITMShortImage(dims, true, false);
        if (fread(rgbs[f]->GetData(MEMORYDEVICE_CPU), 4, P, in) != (size_t)P) return 3;
        if (fread(depths[f]->GetData(MEMORYDEVICE_CPU), 2, P, in) != (size_t)P) return 3;
        float c2w[16];
        if (fread(c2w, 4, 16, in) != 16) return 3;
        // row-major 4x4 -> ORUtils column-major storage (cv_utils tensorToInfiMatrix4 does the same transpose)
        ORUtils::Matrix4<float> *m = new ORUtils::Matrix4<float>();
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) m->m[c * 4 + r] = c2w[r * 4 + c];
        poses[f] = m;
    }
    struct Free { int32_t frame; float c2w[16]; };
    std::vector<Free> frees(h.nfree);
    for (int k = 0; k < h.nfree; k++) {
        if (fread(&frees[k].frame, 4, 1, in) != 1) return 3;
        if (fread(frees[k].c2w, 4, 16, in) != 16) return 3;
    }
    fclose(in);
    eng->gtC2wPoses = poses;

    if (mode == "time" || mode == "timetrack") {   // timetrack: the same loop with the depth tracker estimating every pose
        // CLIEngine::ProcessFrame loop of the TSDF-only `recon` mode (slam/TsdfFusion/CLIEngine.cpp:34-58); the first
        // frame (bulk allocation) is excluded, as BASELINE.md prescribes
        struct timespec t0, t1;
        eng->ProcessFrame(rgbs[0], depths[0]);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int f = 1; f < h.nframes; f++) eng->ProcessFrame(rgbs[f], depths[f]);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        int threads = 1;
#ifdef _OPENMP
        threads = omp_get_max_threads();
#endif
        printf("{\"frames\": %d, \"seconds\": %.6f, \"threads\": %d, \"tracking\": %d}\n", h.nframes - 1, sec, threads, track_mode ? 1 : 0);
        return 0;
    }

    for (int f = 0; f < h.nframes; f++) {
        eng->ProcessFrame(rgbs[f], depths[f]);
        ITMTrackingState *ts = eng->GetTrackingState();
        ORUtils::Matrix4<float> M = ts->pose_d->GetM(), invM = ts->pose_d->GetInvM();
        chunk("M", f, M.m, 64);
        chunk("invM", f, invM.m, 64);
        if (track_mode) { float sc[2] = {ts->trackerScore, (float)ts->framesProcessed}; chunk("trk_score", f, sc, 8); }
        ITMRenderState_VH *rs = (ITMRenderState_VH *)eng->renderState_live;
        ITMScene<ITMVoxel, ITMVoxelIndex> *scene = eng->GetScene();
        int32_t counts[3] = {rs->noVisibleEntries, scene->localVBA.lastFreeBlockId, scene->index.GetLastFreeExcessListId()};
        chunk("counts", f, counts, 12);
        chunk("visible_ids", f, rs->GetVisibleEntryIDs(), (int64_t)rs->noVisibleEntries * 4);
        {
            const unsigned char *vt = rs->GetEntriesVisibleType();
            std::vector<int32_t> nz;
            for (int i = 0; i < ITMVoxelBlockHash::noTotalEntries; i++)
                if (vt[i]) { nz.push_back(i); nz.push_back(vt[i]); }
            chunk("vis_type_nz", f, nz.data(), (int64_t)nz.size() * 4);
        }
        bool full = (h.dump_vba_every > 0 && ((f + 1) % h.dump_vba_every == 0)) || f == h.nframes - 1;
        dump_scene(eng, f, full);
        chunk("depth_f", f, eng->GetView()->depth->GetData(MEMORYDEVICE_CPU), (int64_t)P * 4);
        chunk("minmax", f, rs->renderingRangeImage->GetData(MEMORYDEVICE_CPU), (int64_t)P * 8);
        chunk("raycast", f, rs->raycastResult->GetData(MEMORYDEVICE_CPU), (int64_t)P * 16);
        chunk("icp_points", f, ts->pointCloud->locations->GetData(MEMORYDEVICE_CPU), (int64_t)P * 16);
        chunk("icp_normals", f, ts->pointCloud->colours->GetData(MEMORYDEVICE_CPU), (int64_t)P * 16);

        for (int k = 0; k < h.nfree; k++) {
            if (frees[k].frame != f) continue;
            // slam_pipeline.cpp:362-371 path: SE3Pose from the stored pose, then runRaycast(pose, intrinsics)
            ORUtils::Matrix4<float> c2w;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) c2w.m[c * 4 + r] = frees[k].c2w[r * 4 + c];
            ORUtils::SE3Pose pose;
            pose.SetInvM(c2w);
            pose.Coerce();
            ITMIntrinsics intr = calib.intrinsics_d;
            eng->runRaycast(&pose, &intr);
            ITMRenderState_VH *fs = (ITMRenderState_VH *)eng->renderState_freeview;
            ORUtils::Matrix4<float> fM = pose.GetM(), fInv = pose.GetInvM();
            int tag = f * 1000 + k;
            chunk("fv_M", tag, fM.m, 64);
            chunk("fv_invM", tag, fInv.m, 64);
            int32_t nv = fs->noVisibleEntries;
            chunk("fv_counts", tag, &nv, 4);
            chunk("fv_visible_ids", tag, fs->GetVisibleEntryIDs(), (int64_t)nv * 4);
            chunk("fv_minmax", tag, fs->renderingRangeImage->GetData(MEMORYDEVICE_CPU), (int64_t)P * 8);
            chunk("fv_raycast", tag, eng->GetFreeVertex()->GetData(MEMORYDEVICE_CPU), (int64_t)P * 16);
            chunk("fv_colour", tag, eng->GetFreeImage()->GetData(MEMORYDEVICE_CPU), (int64_t)P * 4);
        }
    }
    if (argc >= 4 && std::string(argv[3]) == "mesh") dump_mesh(eng->GetScene(), h.nframes - 1, getenv("GPS_REF_SAVE_DIR"));
    fclose(g_out);
    return 0;
}

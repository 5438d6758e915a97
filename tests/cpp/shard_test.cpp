// shard_test.cpp -- a C++ host shards a hypothesis batch over the GPUs of a node with no Python anywhere:
// one process, one host thread per GPU, pr_comm_init_all, contiguous pose shards (pr_shard_range), no data-path collective,
// ONE pr_gather_results of the 72-byte records to rank 0.  Verified against the same batch refined unsharded on device 0,
// bit for bit.  Also: the same shards as "virtual ranks" run one after the other on device 0 (exercises uneven shard sizes on
// a one-GPU box), and two host threads with private contexts (pr_thread_context) sharing device 0.
//   usage: shard_test <golden dir>/ [n_poses]
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"

using cuda_renderer::Model;

static std::vector<Model::mat4x4> make_poses(size_t n)
{   // hypotheses around the test.cpp scene pose (deterministic; the exact SURVEY 8d stream is what bench.py / synth.py generate)
    std::mt19937 rng(6);
    std::uniform_real_distribution<float> ang(-0.17f, 0.17f), tr(-20.f, 20.f);
    const float R0[9] = { 0.34768538f, 0.93761126f, 0.0f, 0.70540612f, -0.26157897f, -0.65877056f, -0.61767070f, 0.22904489f, -0.75234390f };
    std::vector<Model::mat4x4> out(n);
    for (size_t i = 0; i < n; ++i) {
        const float a = ang(rng), b = ang(rng), c = ang(rng);
        const float Rz[9] = { std::cos(c), -std::sin(c), 0, std::sin(c), std::cos(c), 0, 0, 0, 1 };
        const float Ry[9] = { std::cos(b), 0, std::sin(b), 0, 1, 0, -std::sin(b), 0, std::cos(b) };
        const float Rx[9] = { 1, 0, 0, 0, std::cos(a), -std::sin(a), 0, std::sin(a), std::cos(a) };
        float t1[9], t2[9], R[9];
        auto mm = [](const float *A, const float *B, float *C) { for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) { float s = 0; for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[k * 3 + q]; C[r * 3 + q] = s; } };
        mm(Rz, Ry, t1); mm(t1, Rx, t2); mm(t2, R0, R);
        const float t[3] = { 20.f + tr(rng), 20.f + tr(rng), 320.f + tr(rng) };
        out[i].init_from_ptr(R, t);
    }
    return out;
}

struct Job {
    std::vector<Model::Triangle> tris;
    std::vector<int32_t> scene_depth;
    Model::mat4x4 proj;
    float K[9];
    int W = 640, H = 480;
    pr_criteria crit{ 0.f, 0.f, 8 };
};

// everything one rank does: upload mesh + scene to ITS device, refine its shard, leave the results on the device
static int run_rank(const Job &job, const std::vector<Model::mat4x4> &poses, uint32_t first, uint32_t count, pr_result *results_dev)
{
    cuda_renderer::device_vector_holder<Model::Triangle> tris; tris.upload(job.tris);
    ::device_vector_holder<Vec3f> pcd, nrm;
    Scene_projective scene;
    cv::Mat depth(job.H, job.W, CV_32S, const_cast<int32_t *>(job.scene_depth.data()));
    Mat3x3f K(job.K);
    scene.init_Scene_projective_cuda(depth, K, pcd, nrm);
    pr_scene_proj view = scene.c_view();
    if (count == 0) return PR_OK;
    return pr_refine_batch_dev(reinterpret_cast<const pr_triangle *>(tris.data()), tris.size(), reinterpret_cast<const pr_mat4 *>(poses.data() + first), count,
                               job.W, job.H, reinterpret_cast<const pr_mat4 *>(&job.proj), job.K, PR_SCENE_PROJ, &view, job.crit, results_dev, nullptr);
}

int main(int argc, char **argv)
{
    const std::string prefix = argc > 1 ? argv[1] : "tests/golden/";
    const uint32_t P = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 203;          // not a multiple of the world sizes used below
    const int G = pr_device_count();
    if (G < 1) { std::fprintf(stderr, "no device\n"); return 2; }
    if (pr_init(0) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 2; }
    pr_set_option("solve", PR_SOLVE_DEVICE);

    Job job;
    Model model(prefix + "obj_06.ply");
    job.tris = model.tris;
    const float Kd[9] = { 572.4114f, 0.0f, 325.2611f, 0.0f, 573.57043f, 242.04899f, 0.0f, 0.0f, 1.0f };
    std::memcpy(job.K, Kd, sizeof Kd);
    cv::Mat Kcv(3, 3, CV_32F, job.K);
    job.proj = cuda_renderer::compute_proj(Kcv, job.W, job.H);
    const auto poses = make_poses(P);
    {   // scene = render of hypothesis 0
        std::vector<Model::mat4x4> one(1, poses[0]);
        job.scene_depth = cuda_renderer::render_cuda(job.tris, one, job.W, job.H, job.proj);
    }

    // reference: the whole batch on device 0, unsharded
    std::vector<pr_result> whole(P);
    {
        cuda_renderer::device_vector_holder<pr_result> dev(P);
        if (run_rank(job, poses, 0, P, dev.data()) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 1; }
        whole = dev.download();
    }
    int failures = 0;

    // (1) one host thread per GPU + the RCCL gather
    {
        if (pr_comm_init_all(G) != PR_OK) { std::fprintf(stderr, "pr_comm_init_all: %s\n", pr_last_error()); return 1; }
        std::vector<pr_result> gathered(P);
        std::vector<int> rc((size_t)G, PR_OK);
        std::vector<std::string> err((size_t)G);
        std::vector<std::thread> threads;
        for (int d = 0; d < G; ++d)
            threads.emplace_back([&, d]() {
                auto fail = [&](int code) { rc[(size_t)d] = code; err[(size_t)d] = pr_last_error(); };
                if (pr_set_device(d) != PR_OK) return fail(1);
                uint32_t first = 0, count = 0;
                pr_shard_range(P, (uint32_t)d, (uint32_t)G, &first, &count);
                cuda_renderer::device_vector_holder<pr_result> mine(count ? count : 1), all(d == 0 ? P : 1);
                if (run_rank(job, poses, first, count, mine.data()) != PR_OK) return fail(2);
                if (pr_gather_results(mine.data(), count, P, 0, d == 0 ? all.data() : nullptr) != PR_OK) return fail(3);
                if (pr_sync() != PR_OK) return fail(4);
                if (d == 0 && pr_memcpy_d2h(gathered.data(), all.data(), sizeof(pr_result) * P) != PR_OK) return fail(5);
            });
        for (auto &t : threads) t.join();
        for (int d = 0; d < G; ++d) if (rc[(size_t)d] != PR_OK) { std::fprintf(stderr, "rank %d failed at step %d: %s\n", d, rc[(size_t)d], err[(size_t)d].c_str()); ++failures; }
        if (!failures && std::memcmp(gathered.data(), whole.data(), sizeof(pr_result) * P) != 0) { std::fprintf(stderr, "gathered results differ from the unsharded batch\n"); ++failures; }
        pr_set_device(0);
        pr_comm_destroy();
    }

    // (2) the shards of a 5-rank job as virtual ranks on device 0, placed by pr_shard_range
    {
        std::vector<pr_result> stitched(P);
        uint32_t covered = 0;
        for (uint32_t r = 0; r < 5; ++r) {
            uint32_t first = 0, count = 0;
            pr_shard_range(P, r, 5, &first, &count);
            if (first != covered) { std::fprintf(stderr, "shards are not contiguous\n"); ++failures; }
            covered += count;
            cuda_renderer::device_vector_holder<pr_result> dev(count ? count : 1);
            if (run_rank(job, poses, first, count, dev.data()) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 1; }
            if (count) pr_memcpy_d2h(stitched.data() + first, dev.data(), sizeof(pr_result) * count);
        }
        if (covered != P || std::memcmp(stitched.data(), whole.data(), sizeof(pr_result) * P) != 0) { std::fprintf(stderr, "virtual-rank shards differ from the unsharded batch\n"); ++failures; }
    }

    // (3) two host threads with private contexts on device 0, each refining half of the batch at the same time
    {
        std::vector<pr_result> halves(P);
        std::vector<int> rc(2, PR_OK);
        std::vector<std::thread> threads;
        for (int h = 0; h < 2; ++h)
            threads.emplace_back([&, h]() {
                if (pr_set_device(0) != PR_OK || pr_thread_context(1) != PR_OK) { rc[(size_t)h] = 1; return; }
                uint32_t first = 0, count = 0;
                pr_shard_range(P, (uint32_t)h, 2, &first, &count);
                {
                    cuda_renderer::device_vector_holder<pr_result> dev(count);
                    if (run_rank(job, poses, first, count, dev.data()) != PR_OK) rc[(size_t)h] = 2;
                    else pr_memcpy_d2h(halves.data() + first, dev.data(), sizeof(pr_result) * count);
                }
                pr_thread_context(0);
            });
        for (auto &t : threads) t.join();
        if (rc[0] != PR_OK || rc[1] != PR_OK || std::memcmp(halves.data(), whole.data(), sizeof(pr_result) * P) != 0) { std::fprintf(stderr, "private-context threads differ from the unsharded batch\n"); ++failures; }
    }

    double fit = 0;
    for (const auto &r : whole) fit += r.fitness;
    std::printf("{\"devices\": %d, \"poses\": %u, \"mean_fitness\": %.6f, \"failures\": %d}\n", G, P, fit / P, failures);
    return failures ? 1 : 0;
}

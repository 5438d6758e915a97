// pcd_scene.h -- ::Node_kdtree, KDTree_cpu, KDTree_cuda, ::Scene_nn (cuda_icp/scene/pcd_scene/pcd_scene.h:5-137).
#pragma once
#include <cfloat>
#include "../common.h"

struct Node_kdtree {                 // 52 bytes, bit-compatible with pr_kdnode
    int parent = -1, child1 = -1, child2 = -1;
    float split_v = 0;
    float bbox[6] = { 0, 0, 0, 0, 0, 0 };
    int split_dim = 0;
    int left = 0, right = 0;
    bool isleaf() const { return child1 < 0 || child2 < 0; }
};
static_assert(sizeof(Node_kdtree) == sizeof(pr_kdnode), "Node_kdtree layout");

class KDTree_cpu {
public:
    std::vector<Vec3f> pcd_buffer, normal_buffer;
    std::vector<Node_kdtree> nodes;
    void build_tree(int max_num_pcd_in_leaf = 10)       // pcd_scene.cpp:45-184
    {
        assert(pcd_buffer.size() > 0 && pcd_buffer.size() == normal_buffer.size());
        nodes.assign(2 * pcd_buffer.size() + 1, Node_kdtree());
        uint32_t n = 0;
        pose_refine_detail::must(pr_kdtree_build(reinterpret_cast<pr_vec3 *>(pcd_buffer.data()), reinterpret_cast<pr_vec3 *>(normal_buffer.data()),
                                                 pcd_buffer.size(), max_num_pcd_in_leaf, reinterpret_cast<pr_kdnode *>(nodes.data()), nodes.size(), &n), "pr_kdtree_build");
        nodes.resize(n);
    }
};
class KDTree_cuda {
public:
    device_vector_holder<Vec3f> pcd_buffer, normal_buffer;
    device_vector_holder<Node_kdtree> nodes;
};

class Scene_nn {
    float max_dist_diff = 0.1f;
    Vec3f *pcd_ptr = nullptr, *normal_ptr = nullptr;
    Node_kdtree *node_ptr = nullptr;
    uint32_t n_points = 0, n_nodes = 0;      // added: the C ABI wants explicit sizes
    float cam_[4] = { 0, 0, 0, 0 }; uint32_t cam_w_ = 0, cam_h_ = 0;   // added: the camera init_Scene_nn_* was given (the reference drops it): lets a bare ICP() index the scene by pixel
    void remember_camera(Mat3x3f &K, int w, int h) { const float *k = K.data(); cam_[0] = k[0]; cam_[1] = k[4]; cam_[2] = k[2]; cam_[3] = k[5]; cam_w_ = (uint32_t)w; cam_h_ = (uint32_t)h; }
public:
    void init_Scene_nn_cpu(cv::Mat &scene_depth, Mat3x3f &scene_K, KDTree_cpu &kdtree)      // pcd_scene.cpp:4-37
    {
        assert(scene_depth.type() == CV_16U || scene_depth.type() == CV_32S);
        const size_t px = (size_t)scene_depth.rows * scene_depth.cols;
        kdtree.pcd_buffer.assign(px, Vec3f()); kdtree.normal_buffer.assign(px, Vec3f()); kdtree.nodes.assign(2 * px + 1, Node_kdtree());
        pose_refine_detail::must(pr_scene_nn_prepare(scene_depth.data, scene_depth.type() == CV_32S, scene_K.data(), scene_depth.cols, scene_depth.rows, 10,
                                                     reinterpret_cast<pr_vec3 *>(kdtree.pcd_buffer.data()), reinterpret_cast<pr_vec3 *>(kdtree.normal_buffer.data()),
                                                     reinterpret_cast<pr_kdnode *>(kdtree.nodes.data()), kdtree.nodes.size(), &n_points, &n_nodes), "pr_scene_nn_prepare");
        kdtree.pcd_buffer.resize(n_points); kdtree.normal_buffer.resize(n_points); kdtree.nodes.resize(n_nodes);
        remember_camera(scene_K, scene_depth.cols, scene_depth.rows);
        pcd_ptr = kdtree.pcd_buffer.data(); normal_ptr = kdtree.normal_buffer.data(); node_ptr = kdtree.nodes.data();
    }
    // pcd_scene.cu:3-20 prepares the scene on the CPU (init_Scene_nn_cpu) and uploads three arrays -- the step the reference's README names as what is
    // left on the CPU.  Here the depth image goes up once and normals, gather and the level-order build run on the device: the SAME arrays, bit for bit
    // (tests/test_scene_prep_gpu.py), in about a millisecond instead of tens.  -DPOSE_REFINE_CPU_SCENE_PREP keeps the reference's route.
    void init_Scene_nn_cuda(cv::Mat &scene_depth, Mat3x3f &scene_K, KDTree_cuda &kdtree)
    {
#ifdef POSE_REFINE_CPU_SCENE_PREP
        KDTree_cpu cpu;
        init_Scene_nn_cpu(scene_depth, scene_K, cpu);
        kdtree.pcd_buffer.upload(cpu.pcd_buffer); kdtree.normal_buffer.upload(cpu.normal_buffer); kdtree.nodes.upload(cpu.nodes);
        pcd_ptr = kdtree.pcd_buffer.data(); normal_ptr = kdtree.normal_buffer.data(); node_ptr = kdtree.nodes.data();
#else
        bool is32 = false;
        device_vector_holder<unsigned char> depth_dev;
        pose_refine_detail::upload_depth(scene_depth, (size_t)scene_depth.cols, (size_t)scene_depth.rows, depth_dev, is32);     // (a pitched Mat row by row)
        if (is32) init_Scene_nn_device(reinterpret_cast<int32_t *>(depth_dev.__gpu_memory), scene_K, scene_depth.cols, scene_depth.rows, kdtree);
        else init_Scene_nn_device(reinterpret_cast<uint16_t *>(depth_dev.__gpu_memory), scene_K, scene_depth.cols, scene_depth.rows, kdtree);
        // (the arrays are allocated for a frame full of valid pixels; their sizes say what they hold, as the reference's uploads do)
        kdtree.pcd_buffer.__size = n_points; kdtree.normal_buffer.__size = n_points; kdtree.nodes.__size = n_nodes;
#endif
    }
    // normals, valid-pixel gather and the level-order kd-tree build on the device (bit-identical tree)
    template <class T>
    void init_Scene_nn_device(T *scene_depth_dev, Mat3x3f &scene_K, int width, int height, KDTree_cuda &kdtree, int max_leaf = 10)
    {
        const size_t px = (size_t)width * height;
        kdtree.pcd_buffer.__malloc(px); kdtree.normal_buffer.__malloc(px); kdtree.nodes.__malloc(2 * px + 1);
        pose_refine_detail::must(pr_scene_nn_prepare_dev(scene_depth_dev, sizeof(T) == 4, scene_K.data(), width, height, max_leaf,
                                                         reinterpret_cast<pr_vec3 *>(kdtree.pcd_buffer.data()), reinterpret_cast<pr_vec3 *>(kdtree.normal_buffer.data()),
                                                         reinterpret_cast<pr_kdnode *>(kdtree.nodes.data()), 2 * px + 1, &n_points, &n_nodes), "pr_scene_nn_prepare_dev");
        remember_camera(scene_K, width, height);
        pcd_ptr = kdtree.pcd_buffer.data(); normal_ptr = kdtree.normal_buffer.data(); node_ptr = kdtree.nodes.data();
    }
    // pcd_scene.h:60-136, host evaluation over host pointers
    void query(const Vec3f &src, Vec3f &dst, Vec3f &nrm, bool &valid) const
    {
        int cur = 0, prev = -1, best_i = 0; bool up = false; float best = FLT_MAX;
        while (cur >= 0) {
            const Node_kdtree &nd = node_ptr[cur];
            float diff = 0;
            if (nd.split_dim == 0) diff = src.x - nd.split_v;
            if (nd.split_dim == 1) diff = src.y - nd.split_v;
            if (nd.split_dim == 2) diff = src.z - nd.split_v;
            const int nearc = diff < 0 ? nd.child1 : nd.child2, farc = diff < 0 ? nd.child2 : nd.child1;
            if (!up) {
                if (nd.isleaf()) {
                    for (int i = nd.left; i < nd.right; ++i) {
                        float d2 = pow2(src.x - pcd_ptr[i].x) + pow2(src.y - pcd_ptr[i].y) + pow2(src.z - pcd_ptr[i].z);
                        if (d2 < best) { best = d2; best_i = i; }
                    }
                    up = true; prev = cur; cur = nd.parent;
                } else { prev = cur; cur = nearc; }
            } else {
                float lb = 0;
                for (int a = 0; a < 3; ++a) { const float s = src[a]; if (s < nd.bbox[2 * a]) lb += pow2(nd.bbox[2 * a] - s); else if (s > nd.bbox[2 * a + 1]) lb += pow2(nd.bbox[2 * a + 1] - s); }
                if (prev == nearc && lb <= best) { prev = cur; cur = farc; up = false; }
                else { prev = cur; cur = nd.parent; }
            }
        }
        valid = best < pow2(max_dist_diff);
        if (valid) { dst = pcd_ptr[best_i]; nrm = normal_ptr[best_i]; }
    }
    pr_scene_nn c_view() const { pr_scene_nn s; s.max_dist_diff = max_dist_diff; s.pcd = reinterpret_cast<const pr_vec3 *>(pcd_ptr); s.normal = reinterpret_cast<const pr_vec3 *>(normal_ptr);
        s.nodes = reinterpret_cast<const pr_kdnode *>(node_ptr); s.n_points = n_points; s.n_nodes = n_nodes;
        s.cam_fx = cam_[0]; s.cam_fy = cam_[1]; s.cam_cx = cam_[2]; s.cam_cy = cam_[3]; s.cam_w = cam_w_; s.cam_h = cam_h_; s.cam_magic = cam_w_ ? PR_SCENE_NN_CAM_MAGIC : 0u; return s; }
};

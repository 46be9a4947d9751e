// make_eigen_fixtures.cpp — pins the CPU oracle's restatements of Eigen WHERE EIGEN EXISTS (the build container has none:
// SURVEY §8c). The reference calls three Eigen decompositions on its hot path:
//   Matrix<float,6,6>::ldlt().solve(b)               motion_estimator.cpp:823, :1054        -> oracle_gn.c  (vo_ref_ldlt6_solve)
//   Matrix<double,Dynamic,Dynamic>::ldlt().solve(B)  sparse_bundle_adjustment.cpp:460, :531 -> oracle_sba.c (vo_ref_ldlt_solve_f64)
//   JacobiSVD<MatrixXf>(M, ComputeFullV) of a 4x4    core/util/triangulate_3d.cpp:120-123   -> oracle_vo.c  (vo_ref_jacobi_svd4)
// This program runs them on seeded inputs and writes inputs + Eigen's outputs as hexadecimal floats:
//     g++ -O2 -march=native -I/usr/include/eigen3 tests/golden/make_eigen_fixtures.cpp -o /tmp/mef && /tmp/mef > tests/golden/eigen_fixtures.txt
// (-O2 -march=native: the reference's flags, core/CMakeLists.txt:9). tests/test_eigen_pin.py compares the oracle with that file
// bit for bit and is SKIPPED while the file does not exist. Nothing here reads /root/reference.
#include <Eigen/Dense>
#include <cstdint>
#include <cstdio>

static uint64_t g_s = 0x243F6A8885A308D3ull;
static double rnd() {  // splitmix64 -> (-1, 1)
  uint64_t z = (g_s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) / 4503599627370496.0 - 1.0;
}

int main() {
  printf("# eigen %d.%d.%d\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
  for (int c = 0; c < 8; ++c) {  // 6x6 float: J^T W J + damping-like diagonal, as the pose-only BA builds it
    Eigen::Matrix<float, 6, 6> A;
    Eigen::Matrix<float, 6, 1> b;
    Eigen::Matrix<float, 6, 12> J;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 12; ++j) J(i, j) = (float)(rnd() * (i < 3 ? 40.0 : 400.0));
    A = J * J.transpose();
    for (int i = 0; i < 6; ++i) {
      A(i, i) += 1e-5f * A(i, i);
      b(i) = (float)(rnd() * 100.0);
    }
    const Eigen::Matrix<float, 6, 1> x = A.ldlt().solve(b);
    printf("ldlt6");
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) printf(" %a", (double)A(i, j));
    for (int i = 0; i < 6; ++i) printf(" %a", (double)b(i));
    for (int i = 0; i < 6; ++i) printf(" %a", (double)x(i));
    printf("\n");
  }
  const int sizes[3] = {3, 12, 42};
  for (int s = 0; s < 3; ++s)
    for (int c = 0; c < 3; ++c) {  // n x n double, one right-hand side (the 3x3 case is solved for the identity by the caller)
      const int n = sizes[s];
      Eigen::MatrixXd G(n, 2 * n), A(n, n), b(n, 1);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < 2 * n; ++j) G(i, j) = rnd() * (1.0 + (i % 6 < 3 ? 30.0 : 300.0));
      A = G * G.transpose();
      for (int i = 0; i < n; ++i) {
        A(i, i) += 1e-5 * A(i, i);
        b(i, 0) = rnd() * 1000.0;
      }
      const Eigen::MatrixXd x = A.ldlt().solve(b);
      printf("ldltd %d", n);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) printf(" %a", A(i, j));
      for (int i = 0; i < n; ++i) printf(" %a", b(i, 0));
      for (int i = 0; i < n; ++i) printf(" %a", x(i, 0));
      printf("\n");
    }
  for (int c = 0; c < 12; ++c) {  // the DLT matrix of triangulate_3d.cpp:104-118 for a random camera pair / pixel pair
    Eigen::MatrixXf M = Eigen::MatrixXf::Zero(4, 4);
    const float fx = 718.856f, fy = 718.856f;
    M(0, 0) = -fx;
    M(1, 1) = -fy;
    M(0, 2) = (float)(rnd() * 600.0);
    M(1, 2) = (float)(rnd() * 180.0);
    for (int j = 0; j < 4; ++j) {
      M(2, j) = (float)(rnd() * (j < 2 ? 700.0 : 600.0));
      M(3, j) = (float)(rnd() * (j < 2 ? 700.0 : 400.0));
    }
    if (c == 0) M(2, 3) = M(3, 3) = 0.0f;  // (a rank-deficient case)
    Eigen::JacobiSVD<Eigen::MatrixXf> svd(M, Eigen::ComputeFullV);
    const Eigen::MatrixXf V = svd.matrixV();
    printf("svd4");
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) printf(" %a", (double)M(i, j));
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) printf(" %a", (double)V(i, j));
    for (int i = 0; i < 4; ++i) printf(" %a", (double)svd.singularValues()(i));
    printf("\n");
  }
  return 0;
}

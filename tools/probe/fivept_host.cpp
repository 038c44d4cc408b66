// Host build of the five-point solver the GPU pose step runs (imp-release_amd/csrc/pose_fivept.h), for the CPU unit test of its algebra:
//   g++ -O2 -std=c++17 tools/probe/fivept_host.cpp -o /tmp/fivept_host ; echo "x0 y0 x1 y1 (5 lines)" | /tmp/fivept_host
// prints the number of solutions and each E (9 values per line).  tests/test_pose.py compares them with oracle/pose_oracle.py five_point.
#include "../../imp-release_amd/csrc/pose_fivept.h"
#include <cstdio>
int main() {
    double x0[5][2], x1[5][2];
    for (;;) {
        for (int i = 0; i < 5; ++i)
            if (scanf("%lf %lf %lf %lf", &x0[i][0], &x0[i][1], &x1[i][0], &x1[i][1]) != 4) return 0;
        double E[10][9];
        fivept::Work w;
        for (int i = 0; i < 5; ++i) { w.pts[i][0] = x0[i][0]; w.pts[i][1] = x0[i][1]; w.pts[i][2] = x1[i][0]; w.pts[i][3] = x1[i][1]; }
        const int n = fivept::five_point<1>(&E[0][0], w, 0);      // a group of one lane = plain sequential code
        printf("%d\n", n);
        for (int k = 0; k < n; ++k) {
            for (int j = 0; j < 9; ++j) printf("%.17g ", E[k][j]);
            printf("\n");
        }
    }
}

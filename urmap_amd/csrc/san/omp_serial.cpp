// omp_serial.cpp -- the OpenMP runtime calls the host side makes, for the build WITHOUT -fopenmp (the TSan build: libgomp is not
// instrumented, so its barriers read as races; the `#pragma omp` loops run serially there)
#include <chrono>
extern "C" {
double omp_get_wtime(void) { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int omp_get_max_threads(void) { return 1; }
int omp_get_thread_num(void) { return 0; }
int omp_get_num_threads(void) { return 1; }
void omp_set_num_threads(int) {}
}

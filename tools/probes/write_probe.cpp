// write_probe: how fast 1.2 GB reach a fresh file under /tmp -- write() from one thread, pwrite() from several, memcpy into a
// shared mapping from several (tools/probes: development aid for the command line tool's writer; build: g++ -O2 -pthread)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/write_probe.bin";
    const size_t total = (size_t)1200 << 20, chunk = (size_t)8 << 20;
    std::vector<char> src(chunk);
    for (size_t i = 0; i < chunk; ++i) src[i] = (char)('A' + i % 23);
    for (int mode = 0; mode < 3; ++mode)
        for (int nt : {1, 2, 4, 8}) {
            if (mode == 0 && nt > 1) continue;
            unlink(path);
            int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
            if (fd < 0) { perror("open"); return 1; }
            const double t0 = now();
            if (mode == 0) {
                for (size_t off = 0; off < total; off += chunk)
                    if (write(fd, src.data(), chunk) != (ssize_t)chunk) { perror("write"); return 1; }
            } else if (mode == 1) {
                std::vector<std::thread> th;
                for (int t = 0; t < nt; ++t)
                    th.emplace_back([&, t] {
                        for (size_t k = (size_t)t; k * chunk < total; k += (size_t)nt)
                            if (pwrite(fd, src.data(), chunk, (off_t)(k * chunk)) != (ssize_t)chunk) perror("pwrite");
                    });
                for (auto& x : th) x.join();
            } else {
                if (ftruncate(fd, (off_t)total)) { perror("ftruncate"); return 1; }
                char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                std::vector<std::thread> th;
                for (int t = 0; t < nt; ++t)
                    th.emplace_back([&, t] {
                        for (size_t k = (size_t)t; k * chunk < total; k += (size_t)nt) memcpy(m + k * chunk, src.data(), chunk);
                    });
                for (auto& x : th) x.join();
                munmap(m, total);
            }
            const double t1 = now();
            close(fd);
            printf("%s threads %d: %.3f s = %.2f GB/s\n", mode == 0 ? "write " : mode == 1 ? "pwrite" : "mmap  ", nt, t1 - t0, total / (t1 - t0) / 1e9);
        }
    unlink(path);
    return 0;
}

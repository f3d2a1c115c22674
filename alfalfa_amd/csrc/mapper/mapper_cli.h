// mapper_cli.h -- entry point of the `alfalfa_amd` command line tool (see mapper_cli.cpp).
#pragma once
#include <functional>
#include <memory>

#include "mapper.h"

namespace alfa {

// How the command line tool obtains the hot path.  The product binds libalfa_hot (alfalfa_main.cpp);
// tests/ bind a checker with the same entry-point signatures to exercise the host logic without a GPU.
struct HotBinding {
    // fills `ops` for `device` (uploads the index); returns 0 on success
    std::function<int(int device, const alfa_host_index* hidx, HotOps* ops)> make_ops;
    std::function<void(HotOps* ops)> release_ops;
    std::function<void(int device)> bind_thread;     // called once in every worker thread
    std::function<const char*()> last_error;
    // optional: the index files of `prefix` straight to `device` (alfa_index_load_to_device): *hidx null -> the host's share
    // (text, names, offsets) is loaded and returned there; non-null -> used as it is (a further GPU).  Fills `ops` like make_ops.
    std::function<int(const char* prefix, int device, alfa_host_index** hidx, HotOps* ops)> load_ops;
    // optional: build the index arrays on `device` from now on (--gpu-index); unset in bindings without a GPU builder
    std::function<int(int device)> index_build_on_device;
};

int mapper_main(int argc, char** argv, const HotBinding& bind);

}  // namespace alfa

// ray_main.cpp — the `ray` executable: same command line as the reference
// (src/main.cu:7-14, src/ui/CommandLineUI.cu:21-101) with the MI355X backend as
// the tracer.  getopt string "t:r:w:hj:c:s:gq:" is accepted unchanged; -g selects
// the GPU backend in the reference and is implied here (the only backend), -q
// (CUDA threads per block) is accepted and ignored, -j is accepted and ignored (JSON
// overrides are outside this backend's scope), -c <one of the six face files> loads the
// cubemap background (CommandLineUI.cu:40-42,66-68).
// Added: -G <device>, -N <GPUs> (devices -G .. -G + N - 1 render tile shards of the frame in one process, tiles gathered
// over RCCL), -S <seed>, -B / -H (build the BVH on the GPU / on the host; default: on the GPU, the same tree).
#include <unistd.h>

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>

#include "tracer.h"

int main(int argc, char **argv) {
    pt::TraceOptions o;
    int size = 512; // TraceUI.h:55
    std::string cubemap;
    int c;
    while ((c = getopt(argc, argv, "t:r:w:hj:c:s:gq:G:N:S:BH")) != -1) {
        switch (c) {
            case 't': o.threads = atoi(optarg); // falls through into -r, CommandLineUI.cu:29-33
            case 'r': o.depth = atoi(optarg); break;
            case 'w': size = atoi(optarg); break;
            case 's': o.samples = atoi(optarg); break;
            case 'G': o.device = atoi(optarg); break;
            case 'N': o.n_devices = atoi(optarg); break;
            case 'S': o.seed = strtoull(optarg, nullptr, 10); break;
            case 'B': o.device_bvh = 1; break;
            case 'H': o.device_bvh = 0; break;
            case 'c': cubemap = optarg; break;   // CommandLineUI.cu:40-42
            case 'j': case 'q': case 'g': break;
            default:
                std::cerr << "usage: " << argv[0] << " [options] [input.ray output.png]\n"
                          << "  -r <#>  recursion level (default 0)\n  -w <#>  image width (default 512)\n"
                          << "  -s <#>  samples per dimension (default 2)\n  -G <#>  HIP device\n  -N <#>  number of GPUs (starting at -G), one process, tiles gathered over RCCL\n  -S <#>  RNG seed\n  -B      build the BVH on the GPU (binned SAH; the default)\n  -H      build the BVH on the host (the same tree, seconds for a million primitives)\n";
                return 1;
        }
    }
    if (optind >= argc - 1) { // CommandLineUI.cu:70-73
        std::cerr << "no input and/or output name." << std::endl;
        return 1;
    }
    const char *rayName = argv[optind], *imgName = argv[optind + 1];
    pt::HipTracer tracer(o);
    if (!tracer.ok()) return 2;
    if (!cubemap.empty()) tracer.loadCubemap(cubemap.c_str()); // CommandLineUI.cu:66-68; a failure is reported and the render goes on without it
    if (!tracer.loadScene(rayName)) { // CommandLineUI.cu:82,96-100
        std::cerr << "Unable to load ray file '" << rayName << "'" << std::endl;
        return 1;
    }
    int width = size;
    int height = (int)std::round(width / tracer.aspectRatio() + 0.5); // CommandLineUI.cu:83-84
    tracer.traceSetup(width, height);
    tracer.traceImage(width, height);
    tracer.waitRender();
    if (!tracer.ok()) return 2;
    unsigned char *buf;
    tracer.getBuffer(buf, width, height);
    if (buf) pt::writeImage(imgName, width, height, buf);
    const mi_stats &st = tracer.stats();
    std::cerr << width << "x" << height << " rays=" << st.rays_cast << " ms=" << st.ms_total << " Mrays/s="
              << (st.ms_total > 0 ? st.rays_cast / st.ms_total / 1e3 : 0) << std::endl;
    return 0;
}

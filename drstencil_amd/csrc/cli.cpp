// cli.cpp -- the `drstencil` command (reference: main.cpp:10-280).
#include <cstdio>
#include <iostream>
#include "generator.hpp"

int main(int argc, char **argv) {
    std::vector<std::string> args;
    for (int i = 1; i < argc; i++) args.push_back(argv[i]);
    drs::GenResult r = drs::generate(args);
    std::cout << r.messages;
    std::cerr << r.notes;          // e.g. "the tuner's configuration ... is used": not on stdout, which is the reference's byte for byte
    if (!r.plan.error.empty() && r.exit_code != 0) std::cerr << "drstencil: " << r.plan.error << std::endl;
    if (r.emitted && !drs::write_text(r.out_name, r.source)) {
        std::cerr << "drstencil: cannot write " << r.out_name << std::endl;
        return 255;
    }
    return r.exit_code;
}

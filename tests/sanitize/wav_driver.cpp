// The command line tool's WAV reader (cli/wav.hpp: untrusted file headers) on every file named on the command line, under
// AddressSanitizer and UndefinedBehaviorSanitizer.  Prints "<probe ok> <read ok> <channels> <frames>" per file.
#include <cstdio>

#include "wav.hpp"

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; i++) {
        wav::Info a, b;
        std::string err;
        std::vector<float> frames;
        const bool p = wav::probe(argv[i], a, err);
        const bool r = wav::read(argv[i], b, frames, err);
        if (r && (b.channels <= 0 || (long long)frames.size() != (long long)b.frames * b.channels)) {
            std::printf("read() succeeded with an inconsistent shape\n");
            return 3;
        }
        double acc = 0.0;
        for (float v : frames) acc += v == v ? (double)v : 0.0;      // touch every sample that was handed out
        std::printf("%d %d %d %lld %s\n", p ? 1 : 0, r ? 1 : 0, r ? b.channels : 0, r ? (long long)b.frames : 0LL, acc == acc ? "" : "!");
    }
    return 0;
}

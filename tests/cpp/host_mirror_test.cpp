// Drives the C++ mirror of the reference's Swift surface (include/syldet.hpp) the way the reference's
// callers do (TrackDetector.swift:62-77, Processor.swift:120-141): load a text configuration, append
// audio in ragged chunks, drain processNewValue, and also run the same audio as one batch.
//   usage: host_mirror_test net.txt samples.f32 out.f32
// Writes: [n_stream_evals][outputs] floats from the streaming API followed by the batch results.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "syldet.hpp"

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    try {
        // error behaviour first: a missing file throws ParseError.unableToOpenPath
        try {
            syldetxx::SyllableDetectorConfig missing("/nonexistent/net.txt");
            return 3;
        } catch (const syldetxx::ParseError &e) {
            if (e.kind != syldetxx::ParseError::unableToOpenPath) return 4;
        }
        syldetxx::SyllableDetectorConfig config(argv[1]);
        std::FILE *f = std::fopen(argv[2], "rb");
        if (!f) return 5;
        std::vector<float> x;
        float buf[4096];
        size_t n;
        while ((n = std::fread(buf, sizeof(float), 4096, f)) > 0) x.insert(x.end(), buf, buf + n);
        std::fclose(f);

        syldetxx::SyllableDetectorBank bank(config, 1);
        syldetxx::SyllableDetector detector(bank, 0);
        std::vector<float> streamed;
        size_t pos = 0, chunk = 1;
        int detected = 0;
        while (pos < x.size()) {
            const size_t m = std::min(chunk, x.size() - pos);
            detector.appendAudioData(x.data() + pos, (int64_t)m);
            pos += m;
            chunk = chunk * 3 % 1777 + 1;
            if (chunk & 1) bank.processAll();                     // the batched consumer step, every other round
            while (detector.processNewValue()) {
                const std::vector<float> o = detector.lastOutputs();
                streamed.insert(streamed.end(), o.begin(), o.end());
                detected += detector.lastDetected() ? 1 : 0;
            }
        }
        std::vector<float> outputs;
        std::vector<uint8_t> flags;
        bank.run(x.data(), (int64_t)x.size(), outputs, flags);
        const std::vector<int64_t> idx = bank.detections(flags.data(), (int64_t)flags.size(), 0.0, 0);
        // ... and the same recording through the one-process sharded bank (a bank over a list of devices: here device 0 listed
        // twice, so the one channel is cut along time into two stretches with a halo): identical bits
        {
            syldetxx::SyllableDetectorShardedBank sharded(config, 1, {0, 0});
            if (sharded.shards() != 2 || sharded.channels() != 1 || sharded.shard(1).parts != 2) return 7;
            sharded.connect();                                     // (a device listed twice: the copy exchange -- nothing to bring up, no RCCL)
            if (sharded.rcclRanks() != 0 || sharded.launcherThreads() != 2) return 9;
            std::vector<float> so;
            std::vector<uint8_t> sf;
            sharded.run(x.data(), (int64_t)x.size(), so, sf);
            if (so != outputs || sf != flags) {
                std::fprintf(stderr, "host_mirror_test: the sharded bank's results differ from the plain bank's\n");
                return 8;
            }
        }
        std::FILE *o = std::fopen(argv[3], "wb");
        if (!o) return 6;
        std::fwrite(streamed.data(), sizeof(float), streamed.size(), o);
        std::fwrite(outputs.data(), sizeof(float), outputs.size(), o);
        std::fclose(o);
        std::printf("%zu %zu %d %zu %lld\n", streamed.size(), outputs.size(), detected, idx.size(), idx.empty() ? -1LL : (long long)idx[0]);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_mirror_test: %s\n", e.what());
        return 1;
    }
}

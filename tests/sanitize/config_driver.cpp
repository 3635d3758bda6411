// Every file named on the command line through the text-format parser (syldet_config_load_text: the restatement of
// SyllableDetectorConfig.init(fromTextFile:), SyllableDetectorConfig.swift:170-277) and, where it parses, through
// SyllableDetector.init's validation (syldet_config_geometry) and a deep copy -- under AddressSanitizer and
// UndefinedBehaviorSanitizer.  Prints "<status> <geometry status>" per file; any sanitizer report aborts the run.
#include <cstdio>

#include "syldet.h"
#include "syldet_internal.hpp"

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; i++) {
        syldet_config_t *cfg = nullptr;
        const int st = syldet_config_load_text(argv[i], &cfg);
        int gs = 1;
        if (st == SYLDET_OK) {
            syldet_geometry_t g;
            gs = syldet_config_geometry(cfg, &g);
            sd::OwnedConfig copy;                       // what syldet_create does with the caller's arrays
            if (copy.assign(*cfg) != SYLDET_OK) gs = -99;
        } else if (cfg) {
            std::printf("a failed parse must not hand out a configuration\n");
            return 3;
        }
        std::printf("%d %d\n", st, gs);
        syldet_config_free(cfg);
    }
    return 0;
}

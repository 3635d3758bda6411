//  Resampler.swift -- replacement for Common/Resampler.swift's ResamplerLinear over libsyldet
//  (bound through the bridging header, see INTEGRATION.md).  Same protocol, same class name and
//  members, so Processor.swift:116-121 and ViewControllerProcessor.swift:247-250 compile unchanged.
//  Not compiled in this repository (no Swift toolchain in the build image).

import Foundation

protocol Resampler {
    func resampleVector(_ data: UnsafePointer<Float>, ofLength numSamples: Int) -> [Float]
}

/// Linear interpolation on the GPU, bit-identical to the original's vDSP_vramp / vDSP_vlint sequence.
class ResamplerLinear: Resampler {
    let samplingRateIn: Double
    let samplingRateOut: Double
    private var handle: OpaquePointer?

    init(fromRate samplingRateIn: Double, toRate samplingRateOut: Double) {
        self.samplingRateIn = samplingRateIn
        self.samplingRateOut = samplingRateOut
        let st = syldet_resampler_create(samplingRateIn, samplingRateOut, 1, 0, &handle)
        if st != 0 {
            fatalError("\(String(cString: syldet_strerror(st))): \(String(cString: syldet_last_error()))")
        }
    }

    deinit {
        syldet_resampler_destroy(handle)
    }

    func resampleVector(_ data: UnsafePointer<Float>, ofLength numSamplesIn: Int) -> [Float] {
        let numSamplesOut = Int(syldet_resampler_count(handle, Int64(numSamplesIn)))
        var ret = [Float](repeating: 0.0, count: numSamplesOut)
        var produced: Int64 = 0
        let st = syldet_resample(handle, data, Int64(numSamplesIn), Int64(numSamplesIn), &ret, Int64(max(numSamplesOut, 1)), &produced)
        if st != 0 {
            fatalError("\(String(cString: syldet_strerror(st))): \(String(cString: syldet_last_error()))")
        }
        return ret
    }

    func resampleArray(_ arr: [Float]) -> [Float] {
        guard !arr.isEmpty else { return [] }
        return arr.withUnsafeBufferPointer { resampleVector($0.baseAddress!, ofLength: $0.count) }
    }
}

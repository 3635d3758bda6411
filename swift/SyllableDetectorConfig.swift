//  SyllableDetectorConfig.swift (libsyldet shim)
//
//  Keeps the stored fields and init(fromTextFile:) throws of Common/SyllableDetectorConfig.swift:11-45,
//  :170-277; the text format is parsed by libsyldet (syldet_config_load_text) and ParseError keeps its
//  four cases (:50-55).  `net` keeps the two members the rest of the project reads (inputs, outputs:
//  Common/SyllableDetector.swift:53-59,70); the network itself lives in the library.
//  Not compiled in this repository (no Swift toolchain in the build image).

import Foundation

/// Owns the C configuration syldet_config_load_text returned; freed with the last copy of the struct that refers to it.
final class SyldetConfigBox {
    let pointer: UnsafeMutablePointer<syldet_config_t>
    init(_ p: UnsafeMutablePointer<syldet_config_t>) { pointer = p }
    deinit { syldet_config_free(pointer) }
}

struct SyllableDetectorConfig {
    enum Scaling { case linear, log, db }
    enum ParseError: Error {
        case unableToOpenPath(String)
        case missingValue(String)
        case invalidValue(String)
        case mismatchedLength(String)
    }
    /// What callers read of the reference's NeuralNet (Common/NeuralNet.swift:232-238)
    struct Net {
        let inputs: Int
        let outputs: Int
    }

    let samplingRate: Double
    let fourierLength: Int
    let windowLength: Int
    let windowOverlap: Int
    let freqRange: (Double, Double)
    let timeRange: Int
    let spectrogramScaling: Scaling
    let thresholds: [Double]
    let net: Net
    private let owned: SyldetConfigBox

    init(fromTextFile path: String) throws {
        var p: UnsafeMutablePointer<syldet_config_t>? = nil
        let st = syldet_config_load_text(path, &p)
        guard st == 0, let c = p else {
            let msg = String(cString: syldet_last_error())
            switch st {
            case Int32(SYLDET_ERR_PARSE_OPEN.rawValue): throw ParseError.unableToOpenPath(path)
            case Int32(SYLDET_ERR_PARSE_MISSING.rawValue): throw ParseError.missingValue(msg)
            case Int32(SYLDET_ERR_PARSE_LENGTH.rawValue): throw ParseError.mismatchedLength(msg)
            default: throw ParseError.invalidValue(msg)
            }
        }
        owned = SyldetConfigBox(c)
        samplingRate = c.pointee.sampling_rate
        fourierLength = Int(c.pointee.fourier_length)
        windowLength = Int(c.pointee.window_length)
        windowOverlap = Int(c.pointee.window_overlap)
        freqRange = (c.pointee.freq_lo, c.pointee.freq_hi)
        timeRange = Int(c.pointee.time_range)
        spectrogramScaling = [Scaling.linear, .log, .db][Int(c.pointee.scaling)]
        thresholds = Array(UnsafeBufferPointer(start: c.pointee.thresholds, count: Int(c.pointee.n_thresholds)))
        let layers = UnsafeBufferPointer(start: c.pointee.layers, count: Int(c.pointee.n_layers))
        net = Net(inputs: Int(layers.first?.inputs ?? 0), outputs: Int(layers.last?.outputs ?? 0))
    }

    func withCStruct<R>(_ body: (UnsafePointer<syldet_config_t>) -> R) -> R { return body(UnsafePointer(owned.pointer)) }
}

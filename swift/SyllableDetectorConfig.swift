//  SyllableDetectorConfig.swift (libsyldet shim)
//
//  Keeps the stored fields and init(fromTextFile:) throws of Common/SyllableDetectorConfig.swift:11-45,
//  :170-277; the text format is parsed by libsyldet (syldet_config_load_text) and ParseError keeps its
//  four cases (:50-55).  Not compiled in this repository.

import Foundation

struct SyllableDetectorConfig {
    enum Scaling { case linear, log, db }
    enum ParseError: Error {
        case unableToOpenPath(String)
        case missingValue(String)
        case invalidValue(String)
        case mismatchedLength(String)
    }

    let samplingRate: Double
    let fourierLength: Int
    let windowLength: Int
    let windowOverlap: Int
    let freqRange: (Double, Double)
    let timeRange: Int
    let spectrogramScaling: Scaling
    let thresholds: [Double]
    private let owned: UnsafeMutablePointer<syldet_config_t>     // freed by the owning class wrapper in real use

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
        owned = c
        samplingRate = c.pointee.sampling_rate
        fourierLength = Int(c.pointee.fourier_length)
        windowLength = Int(c.pointee.window_length)
        windowOverlap = Int(c.pointee.window_overlap)
        freqRange = (c.pointee.freq_lo, c.pointee.freq_hi)
        timeRange = Int(c.pointee.time_range)
        spectrogramScaling = [Scaling.linear, .log, .db][Int(c.pointee.scaling)]
        thresholds = Array(UnsafeBufferPointer(start: c.pointee.thresholds, count: Int(c.pointee.n_thresholds)))
    }

    func withCStruct<R>(_ body: (UnsafePointer<syldet_config_t>) -> R) -> R { return body(UnsafePointer(owned)) }
}

//  SyllableDetector.swift (libsyldet shim)
//
//  Drop-in for Common/SyllableDetector.swift of gardner-lab/syllable-detector-swift: same class
//  name and the members its callers use (Processor.swift:58,120,124,136-141; TrackDetector.swift:33,62,
//  65-77; ViewControllerSimulator.swift:172,292,311,322), with the arithmetic done by libsyldet on an
//  MI355X through the C ABI in include/syldet.h.  The header is imported the way the reference imports
//  TPCircularBuffer.h: one more line in Common/Common-Bridging-Header.h (#include "syldet.h").
//
//  Not compiled in this repository (no Swift toolchain in the build image); INTEGRATION.md shows where
//  it goes.  Every member the reference's callers touch is here: init(config:), config, audioSettings,
//  processSampleBuffer, captureOutput, appendAudioData, processNewValue, lastOutputs, lastDetected, seenSyllable.  One SyllableDetector here owns a one-channel bank; Processor-style callers that hold many
//  channels should create one bank for all of them (see SyllableDetectorBank below) so that a batch of
//  channels is a single kernel launch.

import Foundation
import AVFoundation

final class SyllableDetectorBank {
    let handle: OpaquePointer
    let config: SyllableDetectorConfig
    let geometry: syldet_geometry_t

    init(config: SyllableDetectorConfig, channels: Int, device: Int32 = 0) {
        self.config = config
        var h: OpaquePointer? = nil
        let st = config.withCStruct { syldet_create($0, Int32(channels), device, Int32(SYLDET_ENGINE_AUTO.rawValue), &h) }
        guard st == 0, let hh = h else {
            // the reference calls fatalError for the same conditions (SyllableDetector.swift:47,54,59)
            fatalError(String(cString: syldet_strerror(st)) + ": " + String(cString: syldet_last_error()))
        }
        handle = hh
        var g = syldet_geometry_t()
        syldet_get_geometry(hh, &g)
        geometry = g
    }

    deinit { syldet_destroy(handle) }

    /// Live use: evaluates what every channel has pending in one device round trip; the detectors'
    /// `processNewValue()` then hand the results out (Processor.swift:128-141 calls this first).
    @discardableResult func processAll() -> Int {
        var queued: Int64 = 0
        syldet_process_all(handle, &queued)
        return Int(queued)
    }
}

class SyllableDetector: NSObject, AVCaptureAudioDataOutputSampleBufferDelegate {
    let config: SyllableDetectorConfig
    private let bank: SyllableDetectorBank
    private let channel: Int32

    // the reference's reader settings, unchanged: Float32, non-interleaved, at the network's rate (:19-23; used by
    // TrackDetector.swift:35 and ViewControllerSimulator.swift:176)
    var audioSettings: [String: AnyObject] {
        get {
            return [AVFormatIDKey: NSNumber(value: kAudioFormatLinearPCM), AVLinearPCMBitDepthKey: NSNumber(value: 32), AVLinearPCMIsFloatKey: true as AnyObject, AVLinearPCMIsNonInterleaved: true as AnyObject, AVSampleRateKey: NSNumber(value: config.samplingRate)]
        }
    }

    var lastOutputs: [Float] {
        var out = [Float](repeating: 0.0, count: Int(bank.geometry.outputs))
        syldet_last_outputs(bank.handle, channel, &out)
        return out
    }
    var lastDetected: Bool { return syldet_last_detected(bank.handle, channel) == 1 }

    init(config: SyllableDetectorConfig) {
        self.config = config
        bank = SyllableDetectorBank(config: config, channels: 1)
        channel = 0
        super.init()
    }

    init(bank: SyllableDetectorBank, channel: Int) {
        config = bank.config
        self.bank = bank
        self.channel = Int32(channel)
        super.init()
    }

    func appendAudioData(_ data: UnsafeMutablePointer<Float>, withSamples numSamples: Int) {
        if syldet_append(bank.handle, channel, data, Int64(numSamples)) != 0 {
            fatalError("Insufficient space on buffer.")      // CircularShortTimeFourierTransform.swift:199
        }
    }

    /// Common/SyllableDetector.swift:81-119: the same format checks, then the samples go to this channel's ring
    /// (callers: TrackDetector.swift:62, ViewControllerSimulator.swift:292)
    func processSampleBuffer(_ sampleBuffer: CMSampleBuffer) {
        let numSamples = CMSampleBufferGetNumSamples(sampleBuffer)
        guard 0 < numSamples else { return }
        guard let format = CMSampleBufferGetFormatDescription(sampleBuffer) else { return }
        let audioDescription = CMAudioFormatDescriptionGetStreamBasicDescription(format)
        let isInterleaved = 1 < (audioDescription?[0].mChannelsPerFrame)! && 0 == ((audioDescription?[0].mFormatFlags)! & kAudioFormatFlagIsNonInterleaved)
        let isFloat = 0 < ((audioDescription?[0].mFormatFlags)! & kAudioFormatFlagIsFloat)
        guard audioDescription?[0].mFormatID == kAudioFormatLinearPCM && isFloat && !isInterleaved && audioDescription?[0].mBitsPerChannel == 32 else {
            fatalError("Invalid audio format.")
        }
        guard let audioBuffer = CMSampleBufferGetDataBuffer(sampleBuffer) else { return }
        var lengthAtOffset: Int = 0, totalLength: Int = 0
        var inSamples: UnsafeMutablePointer<Int8>? = nil
        CMBlockBufferGetDataPointer(audioBuffer, 0, &lengthAtOffset, &totalLength, &inSamples)
        inSamples!.withMemoryRebound(to: Float.self, capacity: numSamples) {
            appendAudioData($0, withSamples: numSamples)
        }
    }

    /// :121-124 (AVCaptureAudioDataOutputSampleBufferDelegate)
    func captureOutput(_ captureOutput: AVCaptureOutput, didOutput sampleBuffer: CMSampleBuffer, from connection: AVCaptureConnection) {
        processSampleBuffer(sampleBuffer)
    }

    func processNewValue() -> Bool { return syldet_process_new_value(bank.handle, channel) == 1 }

    func seenSyllable() -> Bool { return syldet_seen_syllable(bank.handle, channel) == 1 }
}

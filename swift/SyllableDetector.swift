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

    /// What a reader must deliver to this detector (the role of Common/SyllableDetector.swift:19-23; read by
    /// TrackDetector.swift:35 and ViewControllerSimulator.swift:176): packed 32-bit float PCM, one plane per channel,
    /// already at the network's sampling rate -- libsyldet's rings take exactly that layout.
    var audioSettings: [String: AnyObject] {
        var wanted = [String: AnyObject]()
        wanted[AVFormatIDKey] = NSNumber(value: kAudioFormatLinearPCM)
        wanted[AVSampleRateKey] = NSNumber(value: config.samplingRate)
        wanted[AVLinearPCMBitDepthKey] = NSNumber(value: MemoryLayout<Float>.size * 8)
        wanted[AVLinearPCMIsFloatKey] = NSNumber(value: true)
        wanted[AVLinearPCMIsNonInterleaved] = NSNumber(value: true)
        return wanted
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

    /// The role of Common/SyllableDetector.swift:81-119 (callers: TrackDetector.swift:62, ViewControllerSimulator.swift:292):
    /// refuse anything but planar 32-bit float PCM the way the reference does (fatalError), then hand the buffer's samples to
    /// this channel's ring.  Written against the audio buffer list rather than the raw block buffer: a sample buffer whose
    /// block is not contiguous is still delivered whole.
    func processSampleBuffer(_ sampleBuffer: CMSampleBuffer) {
        let frames = CMSampleBufferGetNumSamples(sampleBuffer)
        if frames <= 0 { return }
        guard let desc = CMSampleBufferGetFormatDescription(sampleBuffer),
              let asbd = CMAudioFormatDescriptionGetStreamBasicDescription(desc)?.pointee else { return }
        let planar = asbd.mChannelsPerFrame <= 1 || (asbd.mFormatFlags & kAudioFormatFlagIsNonInterleaved) != 0
        let float32 = asbd.mFormatID == kAudioFormatLinearPCM && (asbd.mFormatFlags & kAudioFormatFlagIsFloat) != 0 && asbd.mBitsPerChannel == 32
        if !(planar && float32) { fatalError("Invalid audio format.") }             // :100-102
        var list = AudioBufferList()
        var block: CMBlockBuffer? = nil
        let st = CMSampleBufferGetAudioBufferListWithRetainedBlockBuffer(sampleBuffer, nil, &list, MemoryLayout<AudioBufferList>.size, nil, nil,
                                                                         kCMSampleBufferFlag_AudioBufferList_Assure16ByteAlignment, &block)
        guard st == noErr, let bytes = list.mBuffers.mData else { return }
        let n = min(frames, Int(list.mBuffers.mDataByteSize) / MemoryLayout<Float>.size)
        appendAudioData(bytes.assumingMemoryBound(to: Float.self), withSamples: n)
    }

    /// :121-127 (AVCaptureAudioDataOutputSampleBufferDelegate): ingest, then drain every evaluation that became available
    func captureOutput(_ captureOutput: AVCaptureOutput, didOutput sampleBuffer: CMSampleBuffer, from connection: AVCaptureConnection) {
        processSampleBuffer(sampleBuffer)
        while processNewValue() {}
    }

    func processNewValue() -> Bool { return syldet_process_new_value(bank.handle, channel) == 1 }

    func seenSyllable() -> Bool { return syldet_seen_syllable(bank.handle, channel) == 1 }
}

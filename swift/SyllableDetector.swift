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

/// One bank over several MI355X of this host, one process: the reference is one process that owns every channel
/// (Processor.swift:57-59 builds a detector per channel, one serial queue drains them all :82,:128-141; main.swift:86-89,
/// :126-130 likewise per track).  The library places a sub-bank and a stream on every listed device, splits the channels into
/// contiguous blocks (time-axis ranges with a halo when there are fewer channels than devices), queues every shard's work
/// before waiting on any, and gathers the detection flags with ONE ncclAllGather of their bits per batch (device form).
///
///     let bank = SyllableDetectorShardedBank(config: config, channels: 4096, devices: [0, 1, 2, 3, 4, 5, 6, 7])
///     let (outputs, flags) = bank.run(samples: recording, samplesPerChannel: n)     // [C][n] in, [C][E][outputs] and [C][E] out
final class SyllableDetectorShardedBank {
    let handle: OpaquePointer
    let config: SyllableDetectorConfig
    let channels: Int
    let outputsPerEvaluation: Int

    init(config: SyllableDetectorConfig, channels: Int, devices: [Int32]) {
        self.config = config
        self.channels = channels
        var h: OpaquePointer? = nil
        let st = config.withCStruct { c in
            devices.withUnsafeBufferPointer { d in
                syldet_create_sharded(c, Int32(channels), d.baseAddress, Int32(d.count), Int32(SYLDET_ENGINE_AUTO.rawValue),
                                      Int32(SYLDET_EXCHANGE_RCCL.rawValue), &h)
            }
        }
        guard st == 0, let hh = h else {
            fatalError(String(cString: syldet_strerror(st)) + ": " + String(cString: syldet_last_error()))
        }
        handle = hh
        var g = syldet_geometry_t()
        syldet_get_geometry(syldet_sharded_bank(hh, 0), &g)
        outputsPerEvaluation = Int(g.outputs)
    }

    deinit { syldet_sharded_destroy(handle) }

    /// Brings the flag exchange up now (RCCL communicators over the bank's devices) instead of inside the first batch that
    /// gathers; false where that fails -- the application may then make the bank again for the copy exchange.
    func connect() -> Bool { return syldet_sharded_connect(handle) == 0 }

    func countEvaluations(samplesPerChannel n: Int) -> Int {
        return max(0, Int(syldet_count_evals(syldet_sharded_bank(handle, 0), Int64(n))))
    }

    /// The whole bank in one call on host buffers ([channels][samplesPerChannel], channel-major): every device's copies and
    /// kernels are in flight together; results land in the caller's rows (no collective in this form).
    func run(samples: UnsafePointer<Float>, samplesPerChannel n: Int) -> (outputs: [Float], flags: [UInt8]) {
        let e = countEvaluations(samplesPerChannel: n)
        var outputs = [Float](repeating: 0, count: channels * e * outputsPerEvaluation)
        var flags = [UInt8](repeating: 0, count: channels * e)
        let st = syldet_sharded_run(handle, samples, Int64(n), Int64(n), &outputs, &flags)
        if st != 0 { fatalError(String(cString: syldet_strerror(st)) + ": " + String(cString: syldet_last_error())) }
        return (outputs, flags)
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
    /// this channel's ring.  Like the reference (:104-118) it reads the FIRST plane of the buffer: a non-interleaved buffer with
    /// several channels (the format check lets those through, as the reference's does, and the reader settings do not pin the
    /// channel count) has one plane per channel, plane 0 first.  The audio buffer list is sized for every plane the buffer
    /// holds (a list for one buffer would fail with kCMSampleBufferError_ArrayTooSmall on a stereo file), the block buffer it
    /// retains is kept alive across the append (with Assure16ByteAlignment the data may be a copy that only the block owns),
    /// and a buffer whose samples cannot be reached falls back to the reference's own route, the block's data pointer.
    func processSampleBuffer(_ sampleBuffer: CMSampleBuffer) {
        let frames = CMSampleBufferGetNumSamples(sampleBuffer)
        if frames <= 0 { return }
        guard let desc = CMSampleBufferGetFormatDescription(sampleBuffer),
              let asbd = CMAudioFormatDescriptionGetStreamBasicDescription(desc)?.pointee else { return }
        let planar = asbd.mChannelsPerFrame <= 1 || (asbd.mFormatFlags & kAudioFormatFlagIsNonInterleaved) != 0
        let float32 = asbd.mFormatID == kAudioFormatLinearPCM && (asbd.mFormatFlags & kAudioFormatFlagIsFloat) != 0 && asbd.mBitsPerChannel == 32
        if !(planar && float32) { fatalError("Invalid audio format.") }             // :100-102

        // how large a list this buffer needs (one AudioBuffer per plane)
        var needed = 0
        var st = CMSampleBufferGetAudioBufferListWithRetainedBlockBuffer(sampleBuffer, &needed, nil, 0, nil, nil, 0, nil)
        if st == noErr && needed >= MemoryLayout<AudioBufferList>.size {
            let raw = UnsafeMutableRawPointer.allocate(byteCount: needed, alignment: MemoryLayout<AudioBufferList>.alignment)
            defer { raw.deallocate() }
            let list = raw.bindMemory(to: AudioBufferList.self, capacity: 1)
            var block: CMBlockBuffer? = nil
            st = CMSampleBufferGetAudioBufferListWithRetainedBlockBuffer(sampleBuffer, nil, list, needed, nil, nil,
                                                                         kCMSampleBufferFlag_AudioBufferList_Assure16ByteAlignment, &block)
            if st == noErr, let held = block {
                let planes = UnsafeMutableAudioBufferListPointer(list)
                if let first = planes.first, let bytes = first.mData {
                    let n = min(frames, Int(first.mDataByteSize) / MemoryLayout<Float>.size)
                    withExtendedLifetime(held) {                       // the samples belong to the block: it outlives the append
                        appendAudioData(bytes.assumingMemoryBound(to: Float.self), withSamples: n)
                    }
                    return
                }
            }
        }
        // the reference's route (:104-118): the block buffer's data pointer, plane 0 at its start
        guard let audioBuffer = CMSampleBufferGetDataBuffer(sampleBuffer) else {
            NSLog("SyllableDetector: unable to get the audio buffer (status %d); %d samples dropped", Int(st), frames)
            return
        }
        var lengthAtOffset = 0, totalLength = 0
        var inSamples: UnsafeMutablePointer<Int8>? = nil
        let got = CMBlockBufferGetDataPointer(audioBuffer, 0, &lengthAtOffset, &totalLength, &inSamples)
        guard got == kCMBlockBufferNoErr, let p = inSamples, lengthAtOffset >= frames * MemoryLayout<Float>.size else {
            fatalError("Unable to read the audio buffer (status \(got)).")          // never drop audio silently: a detector that skips buffers reports wrong sample numbers
        }
        withExtendedLifetime(audioBuffer) {
            p.withMemoryRebound(to: Float.self, capacity: frames) { appendAudioData($0, withSamples: frames) }
        }
    }

    /// :121-127 (AVCaptureAudioDataOutputSampleBufferDelegate): ingest, then drain every evaluation that became available
    func captureOutput(_ captureOutput: AVCaptureOutput, didOutput sampleBuffer: CMSampleBuffer, from connection: AVCaptureConnection) {
        processSampleBuffer(sampleBuffer)
        while processNewValue() {}
    }

    func processNewValue() -> Bool { return syldet_process_new_value(bank.handle, channel) == 1 }

    func seenSyllable() -> Bool { return syldet_seen_syllable(bank.handle, channel) == 1 }
}

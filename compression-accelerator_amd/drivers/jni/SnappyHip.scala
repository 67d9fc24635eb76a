// SnappyHip.scala -- the north star's "thin Scala/JVM driver": the reference's compress(src, len, dst) RoCC-command
// surface (tests/compression.c:20-39) over JNI to the C-ABI HIP launcher (include/snappy_hip.h).
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no JDK, scalac or jni.h (probed every round; INTEGRATION.md
// section 4).  Everything the wrapper does is one call into libsnappy_hip.so per method, which is what the GPU tests
// exercise through ctypes; this file and snappy_hip_jni.c are the two sources a maintainer with a JDK builds:
//   scalac SnappyHip.scala
//   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I<repo>/include snappy_hip_jni.c \
//       -L<repo>/compression-accelerator_amd/csrc -lsnappy_hip -o libsnappy_hip_jni.so
package snappyhip

import java.nio.ByteBuffer

object SnappyHip {
  System.loadLibrary("snappy_hip_jni")

  /** 32 + n + n/6: what `dst` must hold (snappy::MaxCompressedLength). */
  @native def maxCompressedLength(n: Long): Long

  /** SET_LENGTH(len); COMPRESS(src, dst) -> rd.  Direct buffers (the accelerator reads memory, not a JVM array).
    * Returns the compressed length, or a negative SZ_* code (tests/compression.c has no error channel; here there is). */
  @native def compress(src: ByteBuffer, len: Int, dst: ByteBuffer): Long

  /** SET_LENGTH(len); UNCOMPRESS(src, dst) -> rd: 0 on success (tests/compression.c:32-39), else a negative SZ_* code. */
  @native def uncompress(src: ByteBuffer, len: Int, dst: ByteBuffer): Long

  /** The hot path for callers that keep data in HBM: nBlocks blocks of 64 KiB at device address dSrc, slots of
    * slotStride bytes at dDst, uint32 lengths at dOutLen, on HIP stream `stream` (0: the null stream).  Asynchronous. */
  @native def compressBlocksDevice(dSrc: Long, nBlocks: Int, dDst: Long, slotStride: Long, dOutLen: Long, stream: Long): Int

  /** Contiguous block range [begin, end) of rank `rank` of `world` (one process per GPU; no data-path collective). */
  @native def shardRange(nBlocks: Long, rank: Int, world: Int): Array[Long]

  /** The test sentence of tests/compression.c:43, round trip (expected: 130 -> 112 bytes, the oracle's and libsnappy 1.1.8's size). */
  def main(args: Array[String]): Unit = {
    val text = "If Tid the thistle sifter lifts his sieve of sifted thistles, then he'll shift the thistles in his sifting sieve to live unsifted."  // tests/compression.c:43 (130 bytes -> 112)
    val raw = text.getBytes("US-ASCII")
    val src = ByteBuffer.allocateDirect(raw.length); src.put(raw); src.flip()
    val dst = ByteBuffer.allocateDirect(maxCompressedLength(raw.length).toInt)
    val n = compress(src, raw.length, dst)
    require(n > 0, s"compress failed: $n")
    val back = ByteBuffer.allocateDirect(raw.length)
    require(uncompress(dst, n.toInt, back) == 0)
    val out = new Array[Byte](raw.length); back.get(out)
    println(s"${raw.length} -> $n bytes, round trip ${java.util.Arrays.equals(raw, out)}")
  }
}

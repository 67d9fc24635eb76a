/* tests/jni_stub/jni.h -- TEST INFRASTRUCTURE, not a JDK header and not part of the product.
 *
 * The build image has no JDK, so compression-accelerator_amd/drivers/jni/snappy_hip_jni.c (the C side of the north
 * star's Scala/JVM driver) could never be compiled, let alone run.  This header declares exactly the part of the JNI
 * the shim touches -- the scalar typedefs, JNIEXPORT/JNICALL, and a JNIEnv whose function table has the four slots the
 * shim calls (GetDirectBufferAddress, GetDirectBufferCapacity, NewLongArray, SetLongArrayRegion) -- with the calling
 * shape of the real one (`(*env)->Fn(env, ...)`), so that tests/jni_stub/jni_driver.c can stand in for a JVM: it fakes
 * direct ByteBuffers and long[] arrays as small structs and calls the shim's Java_snappyhip_SnappyHip_00024_* entry
 * points the way a JVM would.  What this executes: every line of the shim (its bounds, sign and capacity checks, the
 * RoCC command sequence, the device-pointer entry).  What it does not: SnappyHip.scala itself, System.loadLibrary, a
 * real JVM's buffer objects.  Slot ORDER in the table is this file's own; a real jni.h has ~230 slots. */
#ifndef TESTS_JNI_STUB_JNI_H
#define TESTS_JNI_STUB_JNI_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef jint jsize;
typedef void *jobject;
typedef jobject jlongArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
    jlongArray (*NewLongArray)(JNIEnv *env, jsize len);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
};

#endif

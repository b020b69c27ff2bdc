/*
 * jni.h -- a MINIMAL stand-in written for this repository's tests, NOT the JDK's header: just the JNI types and the
 * JNIEnv function-table entries integration/jni/ggs_jni.c uses, declared with the signatures of the JNI specification
 * (chapter 4, "JNI Functions"), so that `gcc -fsyntax-only` can type-check the glue in an image without a JDK
 * (tests/test_jni_binding.py).  The table's layout is NOT the real one -- never link or run against this.
 */
#ifndef GGS_TEST_JNI_STUB_H
#define GGS_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *env, const char *name);
  jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *message);
  jsize (*GetArrayLength)(JNIEnv *env, jarray array);
  jintArray (*NewIntArray)(JNIEnv *env, jsize length);
  jlongArray (*NewLongArray)(JNIEnv *env, jsize length);
  jdoubleArray (*NewDoubleArray)(JNIEnv *env, jsize length);
  jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
  jlong *(*GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
  jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
  void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
  void (*ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
  void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
  void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
  void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
  void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
};
#endif

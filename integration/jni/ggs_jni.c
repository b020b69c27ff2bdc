/*
 * ggs_jni.c -- JNI glue between cc.mallet.topics.GGSNative (the one class that declares native methods;
 * integration/java/cc/mallet/topics/GGSNative.java) and the C-ABI of include/ggs_hip.h.  JNI resolves a native method
 * by its DECLARING class, so every export is Java_cc_mallet_topics_GGSNative_<method>, whichever sampler subclass
 * calls it.  tests/test_jni_binding.py checks declarations against exports (names, JNI types, arity), both ways.
 * SOURCE ONLY (no jni.h in the build image).  Build on a box with a JDK:
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       integration/jni/ggs_jni.c -Lldagroupedgibbssampler_amd/csrc -lggs_hip -o libggs_jni.so
 *
 * Every native method is a flat copy: Java arrays are pinned with Get<Type>ArrayElements /
 * GetPrimitiveArrayCritical for the duration of ONE C-ABI call, as the header promises.  A
 * non-zero return becomes the exception the Java code itself would have thrown
 * (IllegalStateException for GGS:84-85,116-118; IllegalArgumentException for bad arguments).
 */
#include <jni.h>
#include <stdlib.h>
#include "ggs_hip.h"

#define H(h) ((ggs_handle *)(intptr_t)(h))

static void throw_for(JNIEnv *env, ggs_handle *h, int rc) {
  const char *cls = (rc == GGS_ERR_BAD_ARG) ? "java/lang/IllegalArgumentException" : "java/lang/IllegalStateException";
  (*env)->ThrowNew(env, (*env)->FindClass(env, cls), h ? ggs_last_error(h) : "ggs_create failed");
}
/* A pending exception forbids every further JNI call but a handful (the JNI specification, "Exception handling"): a
 * failing call throws and RETURNS -- nothing else of the native method runs. */
#define CHECK(h, call) do { int rc_ = (call); if (rc_) { throw_for(env, (h), rc_); return; } } while (0)
#define CHECK_RET(h, call, ret) do { int rc_ = (call); if (rc_) { throw_for(env, (h), rc_); return (ret); } } while (0)

JNIEXPORT jlong JNICALL Java_cc_mallet_topics_GGSNative_nCreate(JNIEnv *env, jclass c, jint K, jint V,
    jdoubleArray alpha, jdouble beta, jlong seed, jint device, jint flags, jint burnIn, jint thin) {
  ggs_config cfg = {0};
  ggs_handle *h = 0;
  jdouble *a = (*env)->GetDoubleArrayElements(env, alpha, 0);
  cfg.struct_size = (int32_t)sizeof cfg; cfg.num_topics = K; cfg.num_types = V; cfg.device_id = device;
  cfg.alpha = a; cfg.beta = beta; cfg.seed = (uint64_t)seed; cfg.flags = flags; cfg.phi_burn_in = burnIn; cfg.phi_mean_thin = thin;
  int rc = ggs_create(&cfg, &h);
  (*env)->ReleaseDoubleArrayElements(env, alpha, a, JNI_ABORT);
  if (rc) throw_for(env, 0, rc);
  return (jlong)(intptr_t)h;
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nDestroy(JNIEnv *env, jclass c, jlong h) { ggs_destroy(H(h)); }

JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetCorpus(JNIEnv *env, jclass c, jlong h,
    jlongArray docPtr, jintArray tokens, jlong docBase, jlong tokBase) {
  jsize D = (*env)->GetArrayLength(env, docPtr) - 1;
  jlong *p = (*env)->GetLongArrayElements(env, docPtr, 0);
  jint *t = (*env)->GetIntArrayElements(env, tokens, 0);
  int rc = ggs_set_corpus(H(h), D, (const int64_t *)p, (const int32_t *)t, docBase, tokBase);
  (*env)->ReleaseIntArrayElements(env, tokens, t, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, docPtr, p, JNI_ABORT);
  if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetZ(JNIEnv *env, jclass c, jlong h, jintArray z, jboolean redraw) {
  jint *p = (*env)->GetIntArrayElements(env, z, 0);
  int rc = ggs_set_z(H(h), (const int32_t *)p, redraw ? 1 : 0);
  (*env)->ReleaseIntArrayElements(env, z, p, JNI_ABORT);
  if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetIteration(JNIEnv *env, jclass c, jlong h, jint it) { CHECK(H(h), ggs_set_iteration(H(h), it)); }
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSweepBegin(JNIEnv *env, jclass c, jlong h) { CHECK(H(h), ggs_sweep_begin(H(h))); }
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSweepEnd(JNIEnv *env, jclass c, jlong h) { CHECK(H(h), ggs_sweep_end(H(h))); }
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSweepEndAsync(JNIEnv *env, jclass c, jlong h) { CHECK(H(h), ggs_sweep_end_async(H(h))); }
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSynchronize(JNIEnv *env, jclass c, jlong h) { CHECK(H(h), ggs_synchronize(H(h))); }
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSampleZGivenPhi(JNIEnv *env, jclass c, jlong h, jint n) { CHECK(H(h), ggs_sample_z_given_phi(H(h), n)); }

JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGetZ(JNIEnv *env, jclass c, jlong h, jintArray out) {
  jint *p = (*env)->GetIntArrayElements(env, out, 0); int rc = ggs_get_z(H(h), (int32_t *)p);
  (*env)->ReleaseIntArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGetTypeTopicCounts(JNIEnv *env, jclass c, jlong h, jintArray out) {
  jint *p = (*env)->GetIntArrayElements(env, out, 0); int rc = ggs_get_type_topic_counts(H(h), (int32_t *)p);
  (*env)->ReleaseIntArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGetTopicTotals(JNIEnv *env, jclass c, jlong h, jintArray out) {
  jint *p = (*env)->GetIntArrayElements(env, out, 0); int rc = ggs_get_topic_totals(H(h), (int32_t *)p);
  (*env)->ReleaseIntArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc);
}

JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGetPhi(JNIEnv *env, jclass c, jlong h, jdoubleArray out) {
  jdouble *p = (*env)->GetDoubleArrayElements(env, out, 0); int rc = ggs_get_phi(H(h), p);
  (*env)->ReleaseDoubleArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetPhi(JNIEnv *env, jclass c, jlong h, jdoubleArray in) {
  jdouble *p = (*env)->GetDoubleArrayElements(env, in, 0); int rc = ggs_set_phi(H(h), p);
  (*env)->ReleaseDoubleArrayElements(env, in, p, JNI_ABORT); if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT jint JNICALL Java_cc_mallet_topics_GGSNative_nGetPhiMean(JNIEnv *env, jclass c, jlong h, jdoubleArray out) {
  int32_t n = 0; jdouble *p = (*env)->GetDoubleArrayElements(env, out, 0); int rc = ggs_get_phi_mean(H(h), p, &n);
  (*env)->ReleaseDoubleArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc); return n;
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGetTheta(JNIEnv *env, jclass c, jlong h, jlong b, jlong e, jdoubleArray out) {
  jdouble *p = (*env)->GetDoubleArrayElements(env, out, 0); int rc = ggs_get_theta(H(h), b, e, p);
  (*env)->ReleaseDoubleArrayElements(env, out, p, 0); if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT jdoubleArray JNICALL Java_cc_mallet_topics_GGSNative_nGetTimings(JNIEnv *env, jclass c, jlong h) {
  ggs_timings t; jdouble v[5]; jdoubleArray out = (*env)->NewDoubleArray(env, 5);
  if (ggs_get_timings(H(h), &t)) return out;
  v[0] = t.theta_ms; v[1] = t.z_ms; v[2] = t.merge_ms; v[3] = t.phi_ms; v[4] = t.exchange_ms;
  (*env)->SetDoubleArrayRegion(env, out, 0, 5, v);
  return out;
}

/* diagnostics computed on the device-resident state: two doubles cross JNI, no matrices.  ONE C-ABI call per evaluation
 * (a full pass over the corpus and a device-to-host wait; under pcgs the log posterior also redraws theta): it returns
 * {the documents' side of THIS handle, the (replicated) topic side}; a sharded run adds every handle's first to one second. */
static jdoubleArray pair_of(JNIEnv *env, double a, double b) {
  jdouble v[2]; jdoubleArray out = (*env)->NewDoubleArray(env, 2);
  if (!out) return 0;                                  /* OutOfMemoryError is pending */
  v[0] = a; v[1] = b;
  (*env)->SetDoubleArrayRegion(env, out, 0, 2, v);
  return out;
}
JNIEXPORT jdoubleArray JNICALL Java_cc_mallet_topics_GGSNative_nModelLogLikelihood(JNIEnv *env, jclass c, jlong h) {
  double a = 0, b = 0; CHECK_RET(H(h), ggs_model_log_likelihood(H(h), &a, &b), 0); return pair_of(env, a, b);   /* UPLDA:1674-1694, 1701-1747 */
}
JNIEXPORT jdoubleArray JNICALL Java_cc_mallet_topics_GGSNative_nLogPosterior(JNIEnv *env, jclass c, jlong h) {
  double a = 0, b = 0; CHECK_RET(H(h), ggs_log_posterior(H(h), &a, &b), 0); return pair_of(env, a, b);          /* UPLDA:1573-1634 */
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetTestCorpus(JNIEnv *env, jclass c, jlong h,
                                                                                      jlongArray docPtr, jintArray tokens) {
  jsize D = (*env)->GetArrayLength(env, docPtr) - 1;
  jlong *dp = (*env)->GetLongArrayElements(env, docPtr, 0); jint *tk = (*env)->GetIntArrayElements(env, tokens, 0);
  int rc = ggs_set_test_corpus(H(h), D, (const int64_t *)dp, (const int32_t *)tk, 0);           /* MSLDA:918-923 */
  (*env)->ReleaseLongArrayElements(env, docPtr, dp, JNI_ABORT); (*env)->ReleaseIntArrayElements(env, tokens, tk, JNI_ABORT);
  if (rc) throw_for(env, H(h), rc);
}
JNIEXPORT jdouble JNICALL Java_cc_mallet_topics_GGSNative_nHeldOutLogLikelihood(JNIEnv *env, jclass c, jlong h, jint particles) {
  double total = 0; CHECK_RET(H(h), ggs_heldout_log_likelihood(H(h), particles, 0, &total), 0.0); return total;   /* MPE:85-121 */
}

/* ---- one JVM, n GPUs: the group entry points (include/ggs_hip.h, "multi-GPU"); handles travel as a long[] ---------- */
static ggs_handle **handles_of(JNIEnv *env, jlongArray hs, jsize *n, jlong **raw) {
  *n = (*env)->GetArrayLength(env, hs);
  *raw = (*env)->GetLongArrayElements(env, hs, 0);
  ggs_handle **out = (ggs_handle **)malloc(sizeof(ggs_handle *) * (size_t)*n);
  for (jsize i = 0; i < *n; i++) out[i] = H((*raw)[i]);
  return out;
}
JNIEXPORT jlongArray JNICALL Java_cc_mallet_topics_GGSNative_nGroupCreate(JNIEnv *env, jclass c, jint K, jint V,
    jdoubleArray alpha, jdouble beta, jlong seed, jintArray deviceIds, jint flags, jint burnIn, jint thin) {
  ggs_config cfg = {0};
  jsize n = (*env)->GetArrayLength(env, deviceIds);
  jint *dev = (*env)->GetIntArrayElements(env, deviceIds, 0);
  jdouble *a = (*env)->GetDoubleArrayElements(env, alpha, 0);
  ggs_handle **hs = (ggs_handle **)calloc((size_t)n, sizeof(ggs_handle *));
  cfg.struct_size = (int32_t)sizeof cfg; cfg.num_topics = K; cfg.num_types = V; cfg.alpha = a; cfg.beta = beta; cfg.seed = (uint64_t)seed;
  cfg.flags = flags; cfg.phi_burn_in = burnIn; cfg.phi_mean_thin = thin;
  int rc = ggs_group_create(&cfg, n, (const int32_t *)dev, hs);      /* ncclCommInitAll over the listed devices */
  (*env)->ReleaseDoubleArrayElements(env, alpha, a, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, deviceIds, dev, JNI_ABORT);
  jlongArray out = (*env)->NewLongArray(env, n);
  if (rc) { free(hs); throw_for(env, 0, rc); return out; }
  for (jsize i = 0; i < n; i++) { jlong v = (jlong)(intptr_t)hs[i]; (*env)->SetLongArrayRegion(env, out, i, 1, &v); }
  free(hs);
  return out;
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGroupDestroy(JNIEnv *env, jclass c, jlongArray hs) {
  jsize n; jlong *raw; ggs_handle **h = handles_of(env, hs, &n, &raw);
  ggs_group_destroy(h, n);
  free(h); (*env)->ReleaseLongArrayElements(env, hs, raw, JNI_ABORT);
}
/* z: the corpus-wide topic indicators in (document, position) order; shardTokBase[i] .. shardTokBase[i+1] is shard i's slice */
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGroupSetZ(JNIEnv *env, jclass c, jlongArray hs, jintArray z,
    jlongArray shardTokBase, jboolean redraw) {
  jsize n; jlong *raw; ggs_handle **h = handles_of(env, hs, &n, &raw);
  jint *zp = (*env)->GetIntArrayElements(env, z, 0);
  jlong *base = (*env)->GetLongArrayElements(env, shardTokBase, 0);
  const int32_t **parts = (const int32_t **)malloc(sizeof(int32_t *) * (size_t)n);
  for (jsize i = 0; i < n; i++) parts[i] = (const int32_t *)zp + base[i];
  int rc = ggs_group_set_z(h, n, parts, redraw ? 1 : 0);
  free(parts);
  (*env)->ReleaseLongArrayElements(env, shardTokBase, base, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, z, zp, JNI_ABORT);
  if (rc) throw_for(env, h[0], rc);
  free(h); (*env)->ReleaseLongArrayElements(env, hs, raw, JNI_ABORT);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGroupSweep(JNIEnv *env, jclass c, jlongArray hs, jint sweeps) {
  jsize n; jlong *raw; ggs_handle **h = handles_of(env, hs, &n, &raw);
  int rc = ggs_group_sweep(h, n, sweeps);              /* loopOverBatches + updateCounts + samplePhi for every device, collectives grouped */
  if (rc) throw_for(env, h[0], rc);
  free(h); (*env)->ReleaseLongArrayElements(env, hs, raw, JNI_ABORT);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nGroupGatherCounts(JNIEnv *env, jclass c, jlongArray hs) {
  jsize n; jlong *raw; ggs_handle **h = handles_of(env, hs, &n, &raw);
  int rc = ggs_group_gather_counts(h, n);
  if (rc) throw_for(env, h[0], rc);
  free(h); (*env)->ReleaseLongArrayElements(env, hs, raw, JNI_ABORT);
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nSetGlobalTokenCount(JNIEnv *env, jclass c, jlong h, jlong n) {
  CHECK(H(h), ggs_set_global_token_count(H(h), n));
}

/* ---- scheme=collapsed (SerialCollapsedLDA): the seeded start and the serial chain share ONE Randoms(seed) ---------- */
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nInitZJavaLcg(JNIEnv *env, jclass c, jlong h, jint seed) {
  CHECK(H(h), ggs_init_z_java_lcg(H(h), seed));        /* SerialCollapsedLDA.java:789, continued by the sweeps (MSLDA:206) */
  CHECK(H(h), ggs_init_phi(H(h)));
}
JNIEXPORT void JNICALL Java_cc_mallet_topics_GGSNative_nCollapsedSerialSweep(JNIEnv *env, jclass c, jlong h, jint seed, jint sweeps) {
  CHECK(H(h), ggs_collapsed_serial_sweep(H(h), seed, sweeps));   /* SerialCollapsedLDA.java:159-172 */
}

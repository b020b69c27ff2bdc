/*
 * GGSNative -- every native method of the binding, in ONE class, so that each JNI symbol has exactly one name:
 * Java_cc_mallet_topics_GGSNative_<method> (integration/jni/ggs_jni.c).  JNI resolves a native method by the class that
 * DECLARES it; the sampler subclasses (LDAGroupedGibbsSamplerHIP, LDAPartiallyCollapsedGibbsSamplerHIP,
 * SerialCollapsedLDAHIP) and the bridge GGSDevice therefore declare none of their own and call these.
 *
 * One method per C-ABI entry point of include/ggs_hip.h; the comment names it.  tests/test_jni_binding.py checks, without
 * a JDK, that every declaration here has its JNIEXPORT twin with the matching JNI parameter types and vice versa.
 *
 * SOURCE ONLY: the build image has no JDK.
 */
package cc.mallet.topics;

final class GGSNative {
	static { System.loadLibrary("ggs_jni"); }          // libggs_jni.so -> libggs_hip.so
	private GGSNative() { }

	static final int FLAG_PARANOID = 1, FLAG_SAVE_PHI_MEAN = 2, FLAG_PCGS = 4, FLAG_COLLAPSED = 8;   // GGS_FLAG_*

	// ---- one handle = one GPU ------------------------------------------------------------------------------------
	static native long nCreate(int numTopics, int numTypes, double[] alpha, double beta, long seed, int deviceId,
			int flags, int phiBurnIn, int phiMeanThin);                                   // ggs_create
	static native void nDestroy(long h);                                                  // ggs_destroy
	static native void nSetCorpus(long h, long[] docPtr, int[] tokens, long docBase, long tokBase);   // ggs_set_corpus
	static native void nSetZ(long h, int[] z, boolean redrawPhi);                         // ggs_set_z
	static native void nSetIteration(long h, int iteration);                              // ggs_set_iteration
	static native void nSweepBegin(long h);                                               // ggs_sweep_begin
	static native void nSweepEnd(long h);                                                 // ggs_sweep_end
	static native void nSweepEndAsync(long h);                                            // ggs_sweep_end_async: enqueued, not waited for
	static native void nSynchronize(long h);                                              // ggs_synchronize: waits, raises what the async sweeps flagged
	static native void nSampleZGivenPhi(long h, int sweeps);                              // ggs_sample_z_given_phi
	static native void nGetZ(long h, int[] z);                                            // ggs_get_z
	static native void nGetTypeTopicCounts(long h, int[] nwk);                            // ggs_get_type_topic_counts, [V][K]
	static native void nGetTopicTotals(long h, int[] nk);                                 // ggs_get_topic_totals
	static native void nGetPhi(long h, double[] phi);                                     // ggs_get_phi, [K][V]
	static native void nSetPhi(long h, double[] phi);                                     // ggs_set_phi
	static native int nGetPhiMean(long h, double[] phiMean);                              // ggs_get_phi_mean; returns noSampledPhi
	static native void nGetTheta(long h, long docBegin, long docEnd, double[] theta);     // ggs_get_theta
	static native double[] nGetTimings(long h);                                           // ggs_get_timings: theta, z, merge, phi, exchange (ms, cumulative)
	static native double[] nModelLogLikelihood(long h);                                   // ggs_model_log_likelihood: {doc side, topic side} of ONE evaluation
	static native double[] nLogPosterior(long h);                                         // ggs_log_posterior: {doc side, topic side} of ONE evaluation
	static native void nSetTestCorpus(long h, long[] docPtr, int[] tokens);               // ggs_set_test_corpus
	static native double nHeldOutLogLikelihood(long h, int numParticles);                 // ggs_heldout_log_likelihood
	static native void nSetGlobalTokenCount(long h, long n);                              // ggs_set_global_token_count

	// ---- one JVM, n GPUs: the handles travel as a long[] in rank order ---------------------------------------------
	static native long[] nGroupCreate(int numTopics, int numTypes, double[] alpha, double beta, long seed,
			int[] deviceIds, int flags, int phiBurnIn, int phiMeanThin);                  // ggs_group_create
	static native void nGroupDestroy(long[] handles);                                     // ggs_group_destroy
	static native void nGroupSetZ(long[] handles, int[] z, long[] shardTokBase, boolean redrawPhi);   // ggs_group_set_z
	static native void nGroupSweep(long[] handles, int sweeps);                           // ggs_group_sweep
	static native void nGroupGatherCounts(long[] handles);                                // ggs_group_gather_counts

	// ---- scheme=collapsed: the seeded start and the serial chain share ONE java.util.Random stream -------------------
	static native void nInitZJavaLcg(long h, int seed);                                   // ggs_init_z_java_lcg + ggs_init_phi
	static native void nCollapsedSerialSweep(long h, int seed, int sweeps);               // ggs_collapsed_serial_sweep
}

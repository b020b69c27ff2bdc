/*
 * GGSDevice -- the device-resident sampler state behind an UncollapsedParallelLDA subclass (scheme ggs or pcgs), on one
 * GPU or on several GPUs of the node driven from this one JVM.  It owns the ggs_handle(s), flattens the Java corpus
 * into the CSR the C-ABI takes (include/ggs_hip.h), runs one iteration's device work, and copies the state back into
 * the Java FIELDS the reference's diagnostics read (UPLDA:1573-1758 read fields, not getters).
 *
 * Which GPUs: the optional cfg key gpu_devices (a list of HIP device ordinals), read with the reference's own
 * LDAConfiguration.getIntArrayProperty(String, int[]) (LDAConfiguration.java:109); absent -> device 0.
 *   one device    ggs_create / ggs_set_corpus / ggs_set_z;  per iteration ggs_sweep_begin + ggs_sweep_end
 *   n devices     ggs_group_create (ncclCommInitAll); documents split contiguously by the even rule of
 *                 randomscan/document/EvenSplitBatchBuilder.java:30-44, one shard per GPU, every shard told the global
 *                 index of its first document / token so that the Philox streams address what one GPU would address;
 *                 per iteration ONE ggs_group_sweep: the merge of the reference -- updateCounts (UPLDA:1107-1221) and
 *                 the topic batches of samplePhi (GGS:139-171, EvenSplitTopicBatchBuilder.java:28-39) -- is the
 *                 reduce-scatter / Phi-slice draw / all-gather inside the library.  No NCCL code on the Java side.
 * Results are bit-identical between the two (tests/test_native_exchange_gpu.py proves it through the same C-ABI).
 *
 * Package-private on purpose: it reads the protected fields of the model (same package, cc.mallet.topics).
 * SOURCE ONLY: the build image has no JDK; tests/test_jni_binding.py checks every reference member used here against
 * the reference's declarations.
 */
package cc.mallet.topics;

import cc.mallet.types.FeatureSequence;
import cc.mallet.types.InstanceList;

final class GGSDevice {
	private final UncollapsedParallelLDA model;
	private final int schemeFlags;                    // 0 for ggs, GGSNative.FLAG_PCGS for pcgs
	private long[] handles;                           // ggs_handle* per GPU, rank order
	private long[] shardDocBase, shardTokBase;        // n + 1 boundaries
	private int[] flatZ;                              // N, reused for every copy in either direction
	private boolean javaStateStale = false;
	private boolean testSetUploaded = false;
	private boolean sweepsInFlight = false;           // a ggs_sweep_end_async whose errors have not been asked for yet

	GGSDevice(UncollapsedParallelLDA model, int schemeFlags) {
		this.model = model;
		this.schemeFlags = schemeFlags;
	}

	boolean multi() { return handles.length > 1; }

	/** After super.addInstances (alphabet, data, seeded z0: UPLDA:357-456): corpus and z0 to the device(s), counts and
	 *  the initial Phi there (UPLDA:1287-1294).  Java's own initial phi is superseded by the device's. */
	void upload() {
		int[] devices = model.config.getIntArrayProperty("gpu_devices", new int[] { 0 });
		int n = devices.length, D = model.data.size();
		long[] docPtr = new long[D + 1];
		for (int d = 0; d < D; d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) model.data.get(d).instance.getData()).getLength();
		// one Java array holds the corpus (int indices), and one device at most 2^31 - 1 tokens (ggs_set_corpus)
		if (docPtr[D] > Integer.MAX_VALUE)
			throw new IllegalArgumentException("Corpus of " + docPtr[D] + " tokens: more than Integer.MAX_VALUE ("
					+ Integer.MAX_VALUE + ") cannot be flattened into one int[]; split the corpus over several JVMs");
		int N = (int) docPtr[D];
		int[] tokens = new int[N];
		flatZ = new int[N];
		for (int d = 0; d < D; d++) {
			int len = (int) (docPtr[d + 1] - docPtr[d]);
			System.arraycopy(((FeatureSequence) model.data.get(d).instance.getData()).getFeatures(), 0, tokens, (int) docPtr[d], len);
			System.arraycopy(model.data.get(d).topicSequence.getFeatures(), 0, flatZ, (int) docPtr[d], len);
		}
		int flags = schemeFlags | (model.savePhiMeans() ? GGSNative.FLAG_SAVE_PHI_MEAN : 0);
		shardDocBase = new long[n + 1];
		shardTokBase = new long[n + 1];
		for (int r = 0; r < n; r++)                   // EvenSplitBatchBuilder.java:36-43
			shardDocBase[r + 1] = shardDocBase[r] + D / n + (D % n > r ? 1 : 0);
		for (int r = 0; r <= n; r++) shardTokBase[r] = docPtr[(int) shardDocBase[r]];
		if (n == 1) {
			handles = new long[] { GGSNative.nCreate(model.numTopics, model.numTypes, model.alpha, model.beta, model.getStartSeed(),
					devices[0], flags, model.phiBurnIn, model.phiMeanThin) };
			GGSNative.nSetCorpus(handles[0], docPtr, tokens, 0, 0);
			GGSNative.nSetZ(handles[0], flatZ, true);
		} else {
			handles = GGSNative.nGroupCreate(model.numTopics, model.numTypes, model.alpha, model.beta, model.getStartSeed(), devices,
					flags, model.phiBurnIn, model.phiMeanThin);
			for (int r = 0; r < n; r++) {
				int d0 = (int) shardDocBase[r], d1 = (int) shardDocBase[r + 1];
				long[] sub = new long[d1 - d0 + 1];
				for (int d = d0; d <= d1; d++) sub[d - d0] = docPtr[d] - docPtr[d0];
				int[] subTokens = java.util.Arrays.copyOfRange(tokens, (int) docPtr[d0], (int) docPtr[d1]);
				GGSNative.nSetCorpus(handles[r], sub, subTokens, d0, docPtr[d0]);   // doc_base / tok_base: global indices
				GGSNative.nSetGlobalTokenCount(handles[r], N);
			}
			GGSNative.nGroupSetZ(handles, flatZ, shardTokBase, true);   // counts, the start-up exchange, the initial Phi
		}
		javaStateStale = true;
	}

	/** loopOverBatches (UPLDA:1434-1437).  One GPU: theta draw, z draw, the device's counts.  Several: the whole
	 *  iteration, collectives included -- phiStep() has nothing left to do. */
	void zStep() {
		for (long h : handles) GGSNative.nSetIteration(h, model.currentIteration - 1);   // the sweep increments to currentIteration
		if (multi()) GGSNative.nGroupSweep(handles, 1);
		else GGSNative.nSweepBegin(handles[0]);
		javaStateStale = true;
	}

	/** samplePhi (GGS:139-171).  readBack = a diagnostic or getter of this iteration will look at the result: wait for the
	 *  device and raise what the sweep flagged (ggs_sweep_end).  Otherwise the Phi draw is only enqueued
	 *  (ggs_sweep_end_async): the device's error flags are sticky and surface at the next waiting call -- the next
	 *  readBack iteration, syncToJava(), a diagnostic -- and the host round trip per iteration is saved
	 *  (the loop it serves: UPLDA:645-930). */
	void phiStep(boolean readBack) {
		if (multi()) return;                          // ggs_group_sweep did the whole iteration and waited
		if (readBack) { GGSNative.nSweepEnd(handles[0]); sweepsInFlight = false; }
		else { GGSNative.nSweepEndAsync(handles[0]); sweepsInFlight = true; }
	}

	/** Waits for enqueued sweeps and throws what they flagged (the IllegalStateException of GGS:84-85,116-118). */
	private void settle() {
		if (!sweepsInFlight) return;
		GGSNative.nSynchronize(handles[0]);
		sweepsInFlight = false;
	}

	/** setZIndicators (UPLDA:1797-1843) */
	void setZ(int[][] zIndicators) {
		int p = 0;
		for (int[] doc : zIndicators) {
			if (p + doc.length > flatZ.length) break;
			System.arraycopy(doc, 0, flatZ, p, doc.length);
			p += doc.length;
		}
		if (p != flatZ.length)
			throw new IllegalArgumentException("Count does not sum to nr. types! Sumtotal: " + p + " no.types: " + flatZ.length);
		if (multi()) GGSNative.nGroupSetZ(handles, flatZ, shardTokBase, true);
		else GGSNative.nSetZ(handles[0], flatZ, true);
		javaStateStale = true;
	}

	/** sampleZGivenPhi (UPLDA:975-1014); one GPU only (the group entry points have no such call). */
	void sampleZGivenPhi(int iterations) {
		if (multi()) throw new IllegalStateException("sampleZGivenPhi is not available with several gpu_devices");
		GGSNative.nSampleZGivenPhi(handles[0], iterations);
		javaStateStale = true;
	}

	/** setPhi (UPLDA:1897-1902): the device must sample from the Phi the caller set (ggs_set_phi also restarts the
	 *  running phi mean, as `phiMean = new double[..][..]` does). */
	void setPhi(double[][] phi) {
		double[] flat = new double[model.numTopics * model.numTypes];
		for (int k = 0; k < model.numTopics; k++) System.arraycopy(phi[k], 0, flat, k * model.numTypes, model.numTypes);
		for (long h : handles) GGSNative.nSetPhi(h, flat);
		javaStateStale = true;
	}

	/** getPhiMeans (UPLDA:1954-1966): the running sum lives on the device (GGS:193-197 accumulates there). */
	double[][] phiMeans() {
		double[] flat = new double[model.numTopics * model.numTypes];
		int sampled = GGSNative.nGetPhiMean(handles[0], flat);   // phiMean / noSampledPhi, the division of UPLDA:1959-1964 done natively
		if (sampled == 0) return null;
		double[][] result = new double[model.numTopics][model.numTypes];
		for (int k = 0; k < model.numTopics; k++) System.arraycopy(flat, k * model.numTypes, result[k], 0, model.numTypes);
		return result;
	}

	/** modelLogLikelihood (UPLDA:1644-1758) on the device-resident state: the documents' side summed over the shards
	 *  plus one topic side. */
	double modelLogLikelihood() {
		settle();
		if (multi()) GGSNative.nGroupGatherCounts(handles);   // the per-handle call would start a collective one thread cannot complete
		double ll = 0;
		for (int r = 0; r < handles.length; r++) {            // ONE evaluation per handle: {its documents' side, the replicated topic side}
			double[] sides = GGSNative.nModelLogLikelihood(handles[r]);
			ll += sides[0] + (r == 0 ? sides[1] : 0.0);
		}
		return ll;
	}

	/** The body of the private computeLogPosterior (UPLDA:1573-1634), for its call site UPLDA:820-821. */
	double logPosterior() {
		settle();
		double lp = 0;
		for (int r = 0; r < handles.length; r++) {            // one pass over each shard (under pcgs it also redraws the shard's theta)
			double[] sides = GGSNative.nLogPosterior(handles[r]);
			lp += sides[0] + (r == 0 ? sides[1] : 0.0);
		}
		return lp;
	}

	/** addTestInstances (MSLDA:918-923): ids are indices of the shared training alphabet. */
	void uploadTestSet(InstanceList testSet) {
		long[] docPtr = new long[testSet.size() + 1];
		for (int d = 0; d < testSet.size(); d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) testSet.get(d).getData()).getLength();
		int[] tokens = new int[(int) docPtr[testSet.size()]];
		for (int d = 0; d < testSet.size(); d++)
			System.arraycopy(((FeatureSequence) testSet.get(d).getData()).getFeatures(), 0, tokens, (int) docPtr[d],
					(int) (docPtr[d + 1] - docPtr[d]));
		GGSNative.nSetTestCorpus(handles[0], docPtr, tokens);   // the estimate reads corpus-wide counts only: rank 0 evaluates it
		testSetUploaded = true;
	}

	/** evaluator.evaluateLeftToRight(testSet, numParticles, null) (UPLDA:622,841) on the device-resident counts. */
	double heldOutLogLikelihood(int numParticles) {
		if (!testSetUploaded) throw new IllegalStateException("no test set: addTestInstances first");
		settle();
		if (multi()) GGSNative.nGroupGatherCounts(handles);
		return GGSNative.nHeldOutLogLikelihood(handles[0], numParticles);
	}

	/** Copies the device state into the Java fields the diagnostics and getters read. */
	void syncToJava() {
		if (!javaStateStale) return;
		settle();
		if (multi()) GGSNative.nGroupGatherCounts(handles);   // the counts live as topic slices on the GPUs
		int D = model.data.size(), K = model.numTopics, V = model.numTypes;
		for (int r = 0; r < handles.length; r++) {
			int[] z = new int[(int) (shardTokBase[r + 1] - shardTokBase[r])];
			GGSNative.nGetZ(handles[r], z);
			System.arraycopy(z, 0, flatZ, (int) shardTokBase[r], z.length);
		}
		int p = 0;
		for (int d = 0; d < D; d++) {
			int[] z = model.data.get(d).topicSequence.getFeatures();
			System.arraycopy(flatZ, p, z, 0, z.length);
			p += z.length;
		}
		int[] nwk = new int[V * K];
		GGSNative.nGetTypeTopicCounts(handles[0], nwk);       // identical on every rank after the gather
		for (int w = 0; w < V; w++)
			for (int k = 0; k < K; k++) {
				model.typeTopicCounts[w][k] = nwk[w * K + k];
				model.topicTypeCountMapping[k][w] = nwk[w * K + k];
			}
		GGSNative.nGetTopicTotals(handles[0], model.tokensPerTopic);
		double[] flatPhi = new double[K * V];
		GGSNative.nGetPhi(handles[0], flatPhi);               // Phi is replicated: any rank
		for (int k = 0; k < K; k++) System.arraycopy(flatPhi, k * V, model.phi[k], 0, V);
		if ((schemeFlags & GGSNative.FLAG_PCGS) == 0) {      // thetaMatrix rows, GGS:72 (pcgs draws no theta)
			for (int r = 0; r < handles.length; r++) {
				int d0 = (int) shardDocBase[r], d1 = (int) shardDocBase[r + 1];
				double[] th = new double[(d1 - d0) * K];
				GGSNative.nGetTheta(handles[r], 0, d1 - d0, th);
				for (int d = d0; d < d1; d++) {
					model.thetaMatrix[d] = new double[K];
					System.arraycopy(th, (d - d0) * K, model.thetaMatrix[d], 0, K);
				}
			}
		}
		javaStateStale = false;
	}

	/** cumulative device timers of rank 0: theta, z, merge, phi, exchange (ms); feeds the totals printed at UPLDA:931-939 */
	double[] timings() { return GGSNative.nGetTimings(handles[0]); }

	void destroy() {
		if (handles == null) return;
		if (multi()) GGSNative.nGroupDestroy(handles);
		else GGSNative.nDestroy(handles[0]);
		handles = null;
	}
}

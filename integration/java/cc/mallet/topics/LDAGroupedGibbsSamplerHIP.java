/*
 * LDAGroupedGibbsSamplerHIP -- the reference-side binding of libggs_hip.so.
 *
 * SOURCE ONLY: the build image has no JDK / Maven / MALLET jar, so this file has never been
 * compiled.  It shows, against the reference at the surveyed revision, exactly where the
 * native sweep plugs in:
 *   - it extends LDAGroupedGibbsSampler and keeps scheme "ggs" (UncollapsedParallelLDA.sample
 *     branches on the scheme string, UPLDA:143,710-721), selected by an optional cfg key
 *     ggs_backend=hip in tui/ParallelLDA.createModel (ParallelLDA.java:404-408);
 *   - it overrides the three protected hooks of the sweep, UPLDA:660-687:
 *       loopOverBatches()  -> ggs_sweep_begin   (theta draw, z draw, this device's counts)
 *       updateCounts()     -> nothing left to do (the counts were rebuilt on the device)
 *       samplePhi()        -> ggs_sweep_end     (Phi re-draw, phi-mean accumulation)
 *   - Java-side arrays that diagnostics read as FIELDS (typeTopicCounts, topicTypeCountMapping,
 *     tokensPerTopic, phi, thetaMatrix, each document's topicSequence; UPLDA:1573-1758) are
 *     refreshed lazily by syncToJava(), called from postPhi() only when a diagnostic of this
 *     iteration needs them, and always from postSample();
 *   - getPhiMeans() reads the device's running mean (Java's phiMean[][] stays empty) and setPhi()
 *     uploads the matrix, so the two LDASamplerWithPhi methods act on the state that is sampled from.
 */
package cc.mallet.topics;

import cc.mallet.configuration.LDAConfiguration;
import cc.mallet.types.FeatureSequence;
import cc.mallet.types.InstanceList;
import cc.mallet.types.LabelSequence;

public class LDAGroupedGibbsSamplerHIP extends LDAGroupedGibbsSampler {
	private static final long serialVersionUID = 1L;
	static { System.loadLibrary("ggs_jni"); }          // libggs_jni.so -> libggs_hip.so

	private long handle = 0;                          // ggs_handle*
	private int[] flatZ;                              // N, reused for copy-back
	private boolean javaStateStale = false;

	// --- one native method per C-ABI entry point (integration/jni/ggs_jni.c) ---
	private static native long nCreate(int numTopics, int numTypes, double[] alpha, double beta, long seed,
			int deviceId, int flags, int phiBurnIn, int phiMeanThin);
	private static native void nDestroy(long h);
	private static native void nSetCorpus(long h, long[] docPtr, int[] tokens, long docBase, long tokBase);
	private static native void nSetZ(long h, int[] z, boolean redrawPhi);
	private static native void nSetIteration(long h, int iteration);
	private static native void nSweepBegin(long h);
	private static native void nSweepEnd(long h);
	private static native void nSampleZGivenPhi(long h, int sweeps);
	private static native void nGetZ(long h, int[] z);
	private static native void nGetTypeTopicCounts(long h, int[] nwk);    // [V][K]
	private static native void nGetTopicTotals(long h, int[] nk);
	private static native void nGetPhi(long h, double[] phi);             // [K][V]
	private static native void nSetPhi(long h, double[] phi);
	private static native int  nGetPhiMean(long h, double[] phiMean);     // returns noSampledPhi
	private static native void nGetTheta(long h, long docBegin, long docEnd, double[] theta);
	private static native double[] nGetTimings(long h);                    // theta, z, merge, phi (ms, cumulative)
	private static native double nModelLogLikelihood(long h);              // UPLDA:1644-1758 on the device
	private static native double nLogPosterior(long h);                    // UPLDA:1573-1634 on the device
	private static native void nSetTestCorpus(long h, long[] docPtr, int[] tokens);
	private static native double nHeldOutLogLikelihood(long h, int numParticles);   // MarginalProbEstimatorPlain:85-121

	public LDAGroupedGibbsSamplerHIP(LDAConfiguration config) { super(config); }

	@Override
	public void addInstances(InstanceList training) {
		super.addInstances(training);                 // Java: alphabet, data, seeded z0 (UPLDA:398-406), counts
		int D = data.size();
		long[] docPtr = new long[D + 1];
		for (int d = 0; d < D; d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) data.get(d).instance.getData()).getLength();
		int N = (int) docPtr[D];
		int[] tokens = new int[N];
		flatZ = new int[N];
		for (int d = 0; d < D; d++) {
			int[] t = ((FeatureSequence) data.get(d).instance.getData()).getFeatures();
			int[] z = data.get(d).topicSequence.getFeatures();
			int len = (int) (docPtr[d + 1] - docPtr[d]);
			System.arraycopy(t, 0, tokens, (int) docPtr[d], len);
			System.arraycopy(z, 0, flatZ, (int) docPtr[d], len);
		}
		int flags = (savePhiMeans() ? 2 : 0);
		handle = nCreate(numTopics, numTypes, alpha, beta, startSeed, config.getIntProperty("gpu_device", 0), flags,
				phiBurnIn, phiMeanThin);
		nSetCorpus(handle, docPtr, tokens, 0, 0);
		nSetZ(handle, flatZ, true);                   // counts + initial Phi on the device (UPLDA:1287-1294)
		javaStateStale = true;                        // Java's own initial phi is superseded by the device's
	}

	@Override
	public void addTestInstances(InstanceList testSet) {   // MSLDA:918-923; ids are indices of the shared alphabet
		super.addTestInstances(testSet);
		long[] docPtr = new long[testSet.size() + 1];
		for (int d = 0; d < testSet.size(); d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) testSet.get(d).getData()).getLength();
		int[] tokens = new int[(int) docPtr[testSet.size()]];
		for (int d = 0; d < testSet.size(); d++)
			System.arraycopy(((FeatureSequence) testSet.get(d).getData()).getFeatures(), 0, tokens, (int) docPtr[d],
					(int) (docPtr[d + 1] - docPtr[d]));
		nSetTestCorpus(handle, docPtr, tokens);
	}

	/** What sample() logs at UPLDA:622,841 -- call this instead of building a MarginalProbEstimatorPlain over the
	 *  (stale) Java count arrays: evaluateLeftToRight(testSet, numParticles, null) on the device-resident counts. */
	public double heldOutLogLikelihood(int numParticles) { return nHeldOutLogLikelihood(handle, numParticles); }

	@Override
	public double modelLogLikelihood() { return nModelLogLikelihood(handle); }   // UPLDA:1644-1758, no copy-back

	/** Replaces the body of the private computeLogPosterior (UPLDA:1573-1634) at its call site UPLDA:820-821. */
	public double logPosterior() { return nLogPosterior(handle); }

	@Override
	protected void loopOverBatches() {                  // UPLDA:1434-1437
		nSetIteration(handle, currentIteration - 1);  // ggs_sweep_begin increments to currentIteration
		nSweepBegin(handle);
		javaStateStale = true;
	}

	@Override
	protected void updateCounts() { /* rebuilt on the device inside nSweepBegin */ }

	@Override
	protected void samplePhi() {                        // GGS:139-171
		nSweepEnd(handle);
		if (savePhiMeans() && samplePhiThisIteration()) noSampledPhi++;
	}

	@Override
	public void postPhi() {                             // refresh Java fields only if this iteration reads them
		if (config.computeLikelihood() || testSet != null
				|| (config.getStartDiagnostic(LDAConfiguration.START_DIAG_DEFAULT) > 0
					&& currentIteration >= config.getStartDiagnostic(LDAConfiguration.START_DIAG_DEFAULT)))
			syncToJava();
	}

	@Override
	public void postSample() { syncToJava(); super.postSample(); }

	@Override
	public void setZIndicators(int[][] zIndicators) {   // UPLDA:1797-1843
		int p = 0;
		for (int[] doc : zIndicators) { System.arraycopy(doc, 0, flatZ, p, doc.length); p += doc.length; }
		if (p != flatZ.length)
			throw new IllegalArgumentException("Count does not sum to nr. types! Sumtotal: " + p + " no.types: " + flatZ.length);
		nSetZ(handle, flatZ, true);
		javaStateStale = true;
		syncToJava();
	}

	@Override
	public void sampleZGivenPhi(int iterations) {       // UPLDA:975-1014
		nSampleZGivenPhi(handle, iterations);
		javaStateStale = true;
		syncToJava();
	}

	/** Copies the device state into the Java fields the diagnostics and getters read. */
	void syncToJava() {
		if (!javaStateStale) return;
		nGetZ(handle, flatZ);
		int p = 0;
		for (int d = 0; d < data.size(); d++) {
			int[] z = ((LabelSequence) data.get(d).topicSequence).getFeatures();
			System.arraycopy(flatZ, p, z, 0, z.length);
			p += z.length;
		}
		int[] nwk = new int[numTypes * numTopics];
		nGetTypeTopicCounts(handle, nwk);
		for (int w = 0; w < numTypes; w++)
			for (int k = 0; k < numTopics; k++) {
				typeTopicCounts[w][k] = nwk[w * numTopics + k];
				topicTypeCountMapping[k][w] = nwk[w * numTopics + k];
			}
		nGetTopicTotals(handle, tokensPerTopic);
		double[] flatPhi = new double[numTopics * numTypes];
		nGetPhi(handle, flatPhi);
		for (int k = 0; k < numTopics; k++) System.arraycopy(flatPhi, k * numTypes, phi[k], 0, numTypes);
		double[] th = new double[data.size() * numTopics];
		nGetTheta(handle, 0, data.size(), th);
		for (int d = 0; d < data.size(); d++) {
			thetaMatrix[d] = new double[numTopics];
			System.arraycopy(th, d * numTopics, thetaMatrix[d], 0, numTopics);
		}
		javaStateStale = false;
	}

	@Override public double[][] getPhi() { syncToJava(); return phi; }

	/** UPLDA:1954-1966.  The running sum lives on the device (GGS:193-197 accumulates there), so Java's own phiMean[][]
	 *  is never filled: return the device mean, [K][V], or null with the reference's warning when nothing was sampled. */
	@Override
	public double[][] getPhiMeans() {
		double[] flat = new double[numTopics * numTypes];
		int n = nGetPhiMean(handle, flat);            // phiMean / noSampledPhi, the division of UPLDA:1959-1964 done natively
		if (n == 0) {
			logger.warning("No Phi has yet been sampled! getPhiMeans returns 'null'. Ensure that you have correctly configured 'phi_mean_burnin' and 'phi_mean_thin'");
			return null;
		}
		double[][] result = new double[numTopics][numTypes];
		for (int k = 0; k < numTopics; k++) System.arraycopy(flat, k * numTypes, result[k], 0, numTypes);
		return result;
	}

	/** UPLDA:1897-1902: the device must sample from the Phi the caller set, not from its own last draw
	 *  (ggs_set_phi also restarts the running phi mean at zero, as `phiMean = new double[..][..]` does). */
	@Override
	public void setPhi(double[][] phi) {
		super.setPhi(phi);
		double[] flat = new double[numTopics * numTypes];
		for (int k = 0; k < numTopics; k++) System.arraycopy(phi[k], 0, flat, k * numTypes, numTypes);
		nSetPhi(handle, flat);
		javaStateStale = true;                        // z, counts and theta on the Java side are refreshed lazily as before
	}

	@Override
	public void setPhi(double[][] phi, cc.mallet.types.Alphabet dataAlphabet, cc.mallet.types.Alphabet targetAlphabet) {
		super.setPhi(phi, dataAlphabet, targetAlphabet);   // the alphabet checks and ensureConsistentPhi of UPLDA:1913-1920
		setPhi(phi);
	}
	@Override public int[][] getTypeTopicMatrix() { syncToJava(); return super.getTypeTopicMatrix(); }
	@Override public int[] getTopicTotals() { syncToJava(); return super.getTopicTotals(); }
	@Override public int[][] getZIndicators() { syncToJava(); return super.getZIndicators(); }

	@Override
	protected void finalize() { if (handle != 0) { nDestroy(handle); handle = 0; } }
}

/*
 * LDAPartiallyCollapsedGibbsSamplerHIP -- scheme=pcgs with the sweep on MI355X (createModel case "pcgs",
 * tui/ParallelLDA.java:414-416).  The same binding as LDAGroupedGibbsSamplerHIP over the other reference class: the
 * handle is created with GGS_FLAG_PCGS, so loopOverBatches runs the z loop of UPLDA:1466-1544 (theta integrated out,
 * score (n_dk + alpha_k) * phi[k][w], sequential inside a document) instead of GGS:47-132; counts, the Phi draw
 * (LDAPartiallyCollapsedGibbsSampler.java:48-118: beta-smoothed counts, the ggs draw) and the multi-GPU exchange are the
 * same.  The scheme keeps its name, so UncollapsedParallelLDA.sample takes its non-"ggs" branch (UPLDA:710-714): the
 * diagnostic theta is a fresh draw from the document-topic counts, which ggs_log_posterior reproduces on the device
 * from the Philox stream GGS_PURPOSE_THETA.
 *
 * SOURCE ONLY, like the other files of this directory; checked by tests/test_jni_binding.py.
 */
package cc.mallet.topics;

import cc.mallet.configuration.LDAConfiguration;
import cc.mallet.types.InstanceList;

public class LDAPartiallyCollapsedGibbsSamplerHIP extends LDAPartiallyCollapsedGibbsSampler {
	private static final long serialVersionUID = 1L;
	private final GGSDevice device = new GGSDevice(this, GGSNative.FLAG_PCGS);

	public LDAPartiallyCollapsedGibbsSamplerHIP(LDAConfiguration config) { super(config); }

	@Override
	public void addInstances(InstanceList training) {
		super.addInstances(training);                 // Java: alphabet, data, seeded z0 (UPLDA:398-406), counts
		device.upload();
	}

	@Override
	public void addTestInstances(InstanceList testSet) {   // MSLDA:918-923
		super.addTestInstances(testSet);
		device.uploadTestSet(testSet);
	}

	/** What sample() logs at UPLDA:622,841 -- call this instead of building a MarginalProbEstimatorPlain over the
	 *  (stale) Java count arrays. */
	public double heldOutLogLikelihood(int numParticles) { return device.heldOutLogLikelihood(numParticles); }

	@Override
	public double modelLogLikelihood() { return device.modelLogLikelihood(); }   // UPLDA:1644-1758, no copy-back

	/** Replaces the body of the private computeLogPosterior (UPLDA:1573-1634) at its call site UPLDA:820-821. */
	public double logPosterior() { return device.logPosterior(); }

	@Override
	protected void loopOverBatches() { device.zStep(); }                       // UPLDA:1434-1437

	@Override
	protected void updateCounts() { /* rebuilt on the device inside zStep (several GPUs: the reduce-scatter of the count slices) */ }

	@Override
	protected void samplePhi() {                        // LDAPartiallyCollapsedGibbsSampler.java:48-83
		device.phiStep(readsBackThisIteration());     // no diagnostic due: ggs_sweep_end_async, the host does not wait (UPLDA:645-930)
		if (savePhiMeans() && samplePhiThisIteration()) noSampledPhi++;
	}

	/** Does anything of this iteration read the sampler's state on the host?  The diagnostics of UPLDA:829-905 do when
	 *  compute_likelihood is set, a test set is attached or the iteration has reached start_diagnostic. */
	private boolean readsBackThisIteration() {
		int startDiagnostic = config.getStartDiagnostic(LDAConfiguration.START_DIAG_DEFAULT);
		return config.computeLikelihood() || testSet != null || (startDiagnostic > 0 && currentIteration >= startDiagnostic);
	}

	@Override
	public void postPhi() {                             // refresh Java fields only if this iteration reads them
		super.postPhi();
		if (readsBackThisIteration()) device.syncToJava();
	}

	@Override
	public void postSample() { device.syncToJava(); super.postSample(); }

	@Override
	public void setZIndicators(int[][] zIndicators) {   // UPLDA:1797-1843
		device.setZ(zIndicators);
		device.syncToJava();
	}

	@Override
	public void sampleZGivenPhi(int iterations) {       // UPLDA:975-1014
		device.sampleZGivenPhi(iterations);
		device.syncToJava();
	}

	@Override public double[][] getPhi() { device.syncToJava(); return phi; }

	/** UPLDA:1954-1966: the device mean, [K][V], or null with the reference's warning when nothing was sampled. */
	@Override
	public double[][] getPhiMeans() {
		double[][] result = device.phiMeans();
		if (result == null)
			logger.warning("No Phi has yet been sampled! getPhiMeans returns 'null'. Ensure that you have correctly configured 'phi_mean_burnin' and 'phi_mean_thin'");
		return result;
	}

	@Override
	public void setPhi(double[][] phi) {                // UPLDA:1897-1902
		super.setPhi(phi);
		device.setPhi(phi);
	}

	@Override
	public void setPhi(double[][] phi, cc.mallet.types.Alphabet dataAlphabet, cc.mallet.types.Alphabet targetAlphabet) {
		super.setPhi(phi, dataAlphabet, targetAlphabet);   // the alphabet checks and ensureConsistentPhi of UPLDA:1913-1920
		device.setPhi(phi);
	}

	@Override public int[][] getTypeTopicMatrix() { device.syncToJava(); return super.getTypeTopicMatrix(); }
	@Override public int[] getTopicTotals() { device.syncToJava(); return super.getTopicTotals(); }
	@Override public int[][] getZIndicators() { device.syncToJava(); return super.getZIndicators(); }

	/** theta, z, merge, phi, exchange: cumulative device milliseconds, beside the wall-clock totals of UPLDA:931-939 */
	public double[] deviceTimings() { return device.timings(); }

	@Override
	protected void finalize() { device.destroy(); }
}

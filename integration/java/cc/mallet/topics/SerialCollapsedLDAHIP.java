/*
 * SerialCollapsedLDAHIP -- scheme=collapsed with the token loop on MI355X (createModel case "collapsed",
 * tui/ParallelLDA.java:424-428).
 *
 * SerialCollapsedLDA.sample has no protected hook around its loop over documents (SerialCollapsedLDA.java:159-172): the
 * one overridable call inside it is SimpleLDA's sampleTopicsForOneDoc(FeatureSequence, FeatureSequence) (MALLET 2.0.8;
 * the in-tree twin is ModifiedSimpleLDA.java:158-226).  This subclass overrides that: the call for the FIRST document
 * of an iteration runs the whole iteration on the device, the calls for the other documents return at once, and the
 * call for the LAST document copies z and the counts back into the Java fields that the rest of sample() reads
 * (typeTopicCounts, tokensPerTopic, each document's topicSequence: SerialCollapsedLDA.java:191-276,443-552).
 *
 * Two schedules (cfg key collapsed_schedule, read with LDAConfiguration.getIntArrayProperty; absent or 0 = serial):
 *   0  serial    ggs_collapsed_serial_sweep: the reference's own chain -- one pass over all tokens, counts moved in
 *                place, uniforms from the sampler's ONE java.util.Random(seed), which drew the initial topics first
 *                (SerialCollapsedLDA.java:60-65,789; MSLDA:206).  ggs_init_z_java_lcg reproduces that stream on the
 *                host, so the device starts from the same z0 Java drew and continues the same stream: bit-identical to
 *                the restated Java loop (tests/test_collapsed_gpu.py).  One wave: for parity, not throughput.
 *   1  parallel  ggs_sweep with GGS_FLAG_COLLAPSED: the AD-LDA decomposition (ADLDA.java:176-332), one worker per
 *                document against the sweep-start counts; approximate as ADLDA is, +-1 % held-out log likelihood of the
 *                serial chain, device speed.
 *
 * SOURCE ONLY; checked by tests/test_jni_binding.py.  Members of MALLET's SimpleLDA used here (data, numTopics, numTypes,
 * alpha, beta, typeTopicCounts, tokensPerTopic, sampleTopicsForOneDoc) are third-party: not under /root/reference.
 */
package cc.mallet.topics;

import cc.mallet.configuration.LDAConfiguration;
import cc.mallet.types.FeatureSequence;
import cc.mallet.types.InstanceList;

public class SerialCollapsedLDAHIP extends SerialCollapsedLDA {
	private static final long serialVersionUID = 1L;

	private long handle = 0;                          // ggs_handle*
	private int[] flatZ;
	private int docsSeenThisIteration = 0;
	private boolean parallelSchedule = false;

	public SerialCollapsedLDAHIP(LDAConfiguration config) { super(config); }

	@Override
	public void addInstances(InstanceList training) {
		super.addInstances(training);                 // Java: alphabet, seeded z0 from random.nextInt (SerialCollapsedLDA.java:771-800)
		parallelSchedule = config.getIntArrayProperty("collapsed_schedule", new int[] { 0 })[0] == 1;
		int device = config.getIntArrayProperty("gpu_devices", new int[] { 0 })[0];
		int D = data.size();
		long[] docPtr = new long[D + 1];
		for (int d = 0; d < D; d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) data.get(d).instance.getData()).getLength();
		int N = (int) docPtr[D];
		int[] tokens = new int[N];
		flatZ = new int[N];
		for (int d = 0; d < D; d++)
			System.arraycopy(((FeatureSequence) data.get(d).instance.getData()).getFeatures(), 0, tokens, (int) docPtr[d],
					(int) (docPtr[d + 1] - docPtr[d]));
		double[] alphaVector = new double[numTopics];
		java.util.Arrays.fill(alphaVector, alpha);    // SimpleLDA keeps one scalar alpha = alphaSum / numTopics
		handle = GGSNative.nCreate(numTopics, numTypes, alphaVector, beta, getStartSeed(), device, GGSNative.FLAG_COLLAPSED, 0, 0);
		GGSNative.nSetCorpus(handle, docPtr, tokens, 0, 0);
		GGSNative.nInitZJavaLcg(handle, getStartSeed());   // the same Randoms(seed).nextInt(numTopics) stream Java just consumed
		GGSNative.nGetZ(handle, flatZ);
		int p = 0;
		for (int d = 0; d < D; d++) {                  // a JVM-side self-check of the restated LCG: the device's z0 is Java's z0
			int[] z = data.get(d).topicSequence.getFeatures();
			for (int i = 0; i < z.length; i++)
				if (z[i] != flatZ[p++])
					throw new IllegalStateException("device z0 differs from Randoms(seed).nextInt at document " + d);
		}
	}

	@Override
	protected void sampleTopicsForOneDoc(FeatureSequence tokenSequence, FeatureSequence topicSequence) {
		if (docsSeenThisIteration == 0) {
			if (parallelSchedule) {
				GGSNative.nSetIteration(handle, getCurrentIteration() - 1);
				GGSNative.nSweepBegin(handle);
				GGSNative.nSweepEnd(handle);
			} else {
				GGSNative.nCollapsedSerialSweep(handle, getStartSeed(), 1);
			}
		}
		docsSeenThisIteration++;
		if (docsSeenThisIteration == data.size()) {
			docsSeenThisIteration = 0;
			syncToJava();
		}
	}

	@Override
	public void setZIndicators(int[][] zIndicators) {   // SerialCollapsedLDA.java:579-601
		super.setZIndicators(zIndicators);
		int p = 0;
		for (int[] doc : zIndicators) { System.arraycopy(doc, 0, flatZ, p, doc.length); p += doc.length; }
		GGSNative.nSetZ(handle, flatZ, true);         // the serial chain then starts a new Random(seed) stream (include/ggs_hip.h)
	}

	private void syncToJava() {
		GGSNative.nGetZ(handle, flatZ);
		int p = 0;
		for (int d = 0; d < data.size(); d++) {
			int[] z = data.get(d).topicSequence.getFeatures();
			System.arraycopy(flatZ, p, z, 0, z.length);
			p += z.length;
		}
		int[] nwk = new int[numTypes * numTopics];
		GGSNative.nGetTypeTopicCounts(handle, nwk);
		for (int w = 0; w < numTypes; w++) System.arraycopy(nwk, w * numTopics, typeTopicCounts[w], 0, numTopics);
		GGSNative.nGetTopicTotals(handle, tokensPerTopic);
	}

	@Override
	protected void finalize() { if (handle != 0) { GGSNative.nDestroy(handle); handle = 0; } }
}

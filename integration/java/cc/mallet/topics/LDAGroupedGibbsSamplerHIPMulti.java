/*
 * LDAGroupedGibbsSamplerHIPMulti -- scheme=ggs on several GPUs of one node from ONE JVM (the reference's driver is one
 * process): the group entry points of libggs_hip.so (include/ggs_hip.h, "multi-GPU").
 *
 * SOURCE ONLY, like LDAGroupedGibbsSamplerHIP.java (no JDK in the build image).  What it shows:
 *   - documents are split contiguously by the even rule of randomscan/document/EvenSplitBatchBuilder.java:30-44,
 *     one shard per GPU; every shard's handle gets the global index of its first document / token, so the Philox
 *     streams address what one GPU would address;
 *   - ggs_group_create joins the handles with ncclCommInitAll; from then on the merge of the reference --
 *     updateCounts (UPLDA:1107-1221) and the topic batches of samplePhi (GGS:139-171) -- is inside ggs_group_sweep:
 *     reduce-scatter of the counts by topic slice (EvenSplitTopicBatchBuilder.java:28-39, one batch per GPU), the
 *     Phi draw for the GPU's own topics, all-gather of the slices.  No NCCL code on the Java side;
 *   - loopOverBatches + updateCounts + samplePhi become ONE native call per iteration, so the three protected hooks
 *     are overridden as: loopOverBatches -> nGroupSweep, the other two -> nothing.
 * Results are bit-identical to the one-GPU subclass (tests/test_native_exchange_gpu.py proves it through the same C-ABI).
 */
package cc.mallet.topics;

import cc.mallet.configuration.LDAConfiguration;
import cc.mallet.types.FeatureSequence;
import cc.mallet.types.InstanceList;

public class LDAGroupedGibbsSamplerHIPMulti extends LDAGroupedGibbsSampler {
	private static final long serialVersionUID = 1L;
	static { System.loadLibrary("ggs_jni"); }

	private long[] handles;                           // ggs_handle* per GPU, rank order
	private long[] shardDocBase, shardTokBase;        // n + 1 boundaries
	private int[] flatZ;

	private static native long[] nGroupCreate(int numTopics, int numTypes, double[] alpha, double beta, long seed,
			int[] deviceIds, int flags, int phiBurnIn, int phiMeanThin);
	private static native void nGroupDestroy(long[] handles);
	private static native void nGroupSetZ(long[] handles, int[] z, long[] shardTokBase, boolean redrawPhi);
	private static native void nGroupSweep(long[] handles, int sweeps);
	private static native void nGroupGatherCounts(long[] handles);
	private static native void nSetGlobalTokenCount(long h, long n);
	// per-handle calls shared with the one-GPU subclass (integration/jni/ggs_jni.c)
	private static native void nSetCorpus(long h, long[] docPtr, int[] tokens, long docBase, long tokBase);
	private static native void nSetIteration(long h, int iteration);
	private static native void nGetZ(long h, int[] z);
	private static native void nGetTypeTopicCounts(long h, int[] nwk);
	private static native void nGetTopicTotals(long h, int[] nk);
	private static native void nGetPhi(long h, double[] phi);

	public LDAGroupedGibbsSamplerHIPMulti(LDAConfiguration config) { super(config); }

	@Override
	public void addInstances(InstanceList training) {
		super.addInstances(training);                 // Java: alphabet, data, seeded z0 (UPLDA:398-406)
		int[] devices = config.getIntArrayProperty("gpu_devices");   // optional key; default: all visible GPUs
		int n = devices.length, D = data.size();
		shardDocBase = new long[n + 1];
		shardTokBase = new long[n + 1];
		for (int r = 0; r < n; r++)                   // EvenSplitBatchBuilder.java:36-43
			shardDocBase[r + 1] = shardDocBase[r] + D / n + (D % n > r ? 1 : 0);
		long[] docPtr = new long[D + 1];
		for (int d = 0; d < D; d++)
			docPtr[d + 1] = docPtr[d] + ((FeatureSequence) data.get(d).instance.getData()).getLength();
		int N = (int) docPtr[D];
		int[] tokens = new int[N];
		flatZ = new int[N];
		for (int d = 0; d < D; d++) {
			int len = (int) (docPtr[d + 1] - docPtr[d]);
			System.arraycopy(((FeatureSequence) data.get(d).instance.getData()).getFeatures(), 0, tokens, (int) docPtr[d], len);
			System.arraycopy(data.get(d).topicSequence.getFeatures(), 0, flatZ, (int) docPtr[d], len);
		}
		handles = nGroupCreate(numTopics, numTypes, alpha, beta, startSeed, devices, savePhiMeans() ? 2 : 0, phiBurnIn, phiMeanThin);
		for (int r = 0; r < n; r++) {
			int d0 = (int) shardDocBase[r], d1 = (int) shardDocBase[r + 1];
			shardTokBase[r] = docPtr[d0];
			long[] sub = new long[d1 - d0 + 1];
			for (int d = d0; d <= d1; d++) sub[d - d0] = docPtr[d] - docPtr[d0];
			int[] subTokens = java.util.Arrays.copyOfRange(tokens, (int) docPtr[d0], (int) docPtr[d1]);
			nSetCorpus(handles[r], sub, subTokens, d0, docPtr[d0]);   // doc_base / tok_base: global indices
			nSetGlobalTokenCount(handles[r], N);
		}
		shardTokBase[n] = N;
		nGroupSetZ(handles, flatZ, shardTokBase, true);  // counts, the start-up exchange, the initial Phi
	}

	@Override
	protected void loopOverBatches() {                  // UPLDA:1434-1437 -- and updateCounts, and samplePhi
		for (long h : handles) nSetIteration(h, currentIteration - 1);
		nGroupSweep(handles, 1);
		if (savePhiMeans() && samplePhiThisIteration()) noSampledPhi++;
	}
	@Override protected void updateCounts() { /* inside nGroupSweep: the reduce-scatter of the count slices */ }
	@Override protected void samplePhi() { /* inside nGroupSweep: the Phi draw of each GPU's topic slice + the all-gather */ }

	/** Device state into the Java fields the diagnostics read.  The counts live as topic slices on the GPUs: one grouped
	 *  gather first (a getter of a single handle would start a collective that one thread cannot complete). */
	void syncToJava() {
		nGroupGatherCounts(handles);
		for (int r = 0; r < handles.length; r++) {
			int[] z = new int[(int) (shardTokBase[r + 1] - shardTokBase[r])];
			nGetZ(handles[r], z);
			System.arraycopy(z, 0, flatZ, (int) shardTokBase[r], z.length);
		}
		int p = 0;
		for (int d = 0; d < data.size(); d++) {
			int[] z = data.get(d).topicSequence.getFeatures();
			System.arraycopy(flatZ, p, z, 0, z.length);
			p += z.length;
		}
		int[] nwk = new int[numTypes * numTopics];
		nGetTypeTopicCounts(handles[0], nwk);                        // identical on every rank after the gather
		for (int w = 0; w < numTypes; w++)
			for (int k = 0; k < numTopics; k++) {
				typeTopicCounts[w][k] = nwk[w * numTopics + k];
				topicTypeCountMapping[k][w] = nwk[w * numTopics + k];
			}
		nGetTopicTotals(handles[0], tokensPerTopic);
		double[] flatPhi = new double[numTopics * numTypes];
		nGetPhi(handles[0], flatPhi);                                // Phi is replicated: any rank
		for (int k = 0; k < numTopics; k++) System.arraycopy(flatPhi, k * numTypes, phi[k], 0, numTypes);
	}

	@Override public void postSample() { syncToJava(); super.postSample(); }
	@Override protected void finalize() { if (handles != null) { nGroupDestroy(handles); handles = null; } }
}

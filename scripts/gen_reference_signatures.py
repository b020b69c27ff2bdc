"""Declarations of the reference classes the Java binding (integration/java) extends, overrides or calls ->
tests/golden/reference_signatures.json.  DATA, not source: per class its superclass, and per member the name, modifiers
and parameter / field types.  Run here (where /root/reference exists); tests/test_jni_binding.py reads the fixture, and
re-derives it to detect drift when the reference is present.

    python scripts/gen_reference_signatures.py [--check]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import javadecl  # noqa: E402

REF = "/root/reference/src/main/java/cc/mallet"
FILES = [
    "topics/ModifiedSimpleLDA.java", "topics/UncollapsedParallelLDA.java", "topics/LDAGroupedGibbsSampler.java",
    "topics/LDAPartiallyCollapsedGibbsSampler.java", "topics/SerialCollapsedLDA.java", "topics/LDAGibbsSampler.java",
    "topics/LDASamplerWithPhi.java", "topics/AbortableSampler.java", "configuration/LDAConfiguration.java",
    "configuration/Configuration.java", "topics/tui/ParallelLDA.java",
]
OUT = os.path.join(HERE, "..", "tests", "golden", "reference_signatures.json")

# MALLET 2.0.8 (pom.xml:130-134) is a Maven dependency, not under /root/reference: the members of its classes that the
# binding touches, from the published 2.0.8 API.  ASSUMED, not derived -- the test reports them as such.
THIRD_PARTY = {
    "SimpleLDA": {
        "extends": [], "third_party": "cc.mallet:mallet:2.0.8 cc.mallet.topics.SimpleLDA",
        "fields": {n: {"mods": ["protected"], "type": t} for n, t in {
            "data": "ArrayList<TopicAssignment>", "alphabet": "Alphabet", "topicAlphabet": "LabelAlphabet", "numTopics": "int",
            "numTypes": "int", "alpha": "double", "alphaSum": "double", "beta": "double", "betaSum": "double",
            "oneDocTopicCounts": "int[]", "typeTopicCounts": "int[][]", "tokensPerTopic": "int[]", "random": "Randoms"}.items()},
        "methods": [
            {"name": "sampleTopicsForOneDoc", "mods": ["protected"], "ret": "void", "params": ["FeatureSequence", "FeatureSequence"]},
            {"name": "addInstances", "mods": ["public"], "ret": "void", "params": ["InstanceList"]},
            {"name": "setRandomSeed", "mods": ["public"], "ret": "void", "params": ["int"]},
            {"name": "modelLogLikelihood", "mods": ["public"], "ret": "double", "params": []},
        ]},
    "Object": {"extends": [], "third_party": "java.lang.Object", "fields": {},
               "methods": [{"name": "finalize", "mods": ["protected"], "ret": "void", "params": []}]},
}


def build():
    out = {}
    for rel in FILES:
        with open(os.path.join(REF, rel), encoding="utf-8", errors="replace") as f:
            for t in javadecl.parse_types(f.read()):
                for m in t["methods"]:
                    m.pop("body", None)
                    m.pop("pnames", None)
                t["source"] = "src/main/java/cc/mallet/" + rel
                out[t["name"]] = t
    for name, t in THIRD_PARTY.items():
        out[name] = dict(t, name=name, kind="class", implements=[])
    return out


if __name__ == "__main__":
    sig = build()
    text = json.dumps(sig, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("%d classes -> %s" % (len(sig), os.path.relpath(OUT)))

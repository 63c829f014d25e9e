"""Writes tests/golden/ZAM_Tjunction-1_42_T-1.scenario.json: the numbers this package's stdlib reader extracts from the
reference's example scenario (lanelet bounds and topology, obstacle shapes and recorded states, the planning problem) in
the compact JSON form of commonroad_xml.scenario_to_dict.  Data only -- BASELINE configs 1 and 4 run on it on the GPU
box, where /root/reference does not exist.

    python tests/golden/gen_scenario_fixture.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from frenetix_motion_planner_amd import commonroad_xml as crx  # noqa: E402

SRC = "/root/reference/example_scenarios/ZAM_Tjunction-1_42_T-1.xml"

if __name__ == "__main__":
    sc = crx.read_scenario(SRC)
    out = os.path.join(HERE, "ZAM_Tjunction-1_42_T-1.scenario.json")
    crx.write_scenario_json(sc, out)
    back = crx.read_scenario_json(out)
    assert crx.scenario_to_dict(back) == crx.scenario_to_dict(sc)
    print(out, os.path.getsize(out), "bytes;", len(sc.lanelets), "lanelets,", len(sc.obstacles), "obstacles")
